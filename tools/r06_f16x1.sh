#!/bin/bash
# round 6: the opt-in single-pass f16 mode (DDPO_MX_CROSS=0) — kernel contract, parity margins at size (recorded, not gated), sampling A/B against the shipped operator
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_f16x1.log
rm -f gpurun_out/r06_parity_f16x1.log
{
echo "== kernel contract"; timeout 600 python -m pytest tests/test_gpu_f16mx.py -k "single_pass_f16 or tall_tile" -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo "== parity margins under DDPO_MX_CROSS=0 (the suite's gates are the shipped operator's: failures of LP budgets are expected and only the records matter)"
DDPO_MX_CROSS=0 DDPO_TRAJ_STEPS=12 DDPO_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r06_parity_f16x1.log timeout 2400 python -m pytest tests/test_gpu_headline_geometry.py tests/test_gpu_train_parity.py -k "headline or sd15_full_size" -m gpu -q -p no:cacheprovider 2>&1 | tail -12
DDPO_MX_CROSS=0 timeout 600 python -m pytest tests/test_gpu_f16mx_model.py -k "accuracy or ratio_is_one" -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -12
cat gpurun_out/r06_parity_f16x1.log
ENVS="DDPO_MX_CROSS=1;DDPO_MX_CROSS=0" ROUNDS=2 LOG=r06_ab_f16x1.log bash tools/ab_bench.sh
} > $L 2>&1
tail -30 $L | cut -c1-400
