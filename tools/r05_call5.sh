#!/bin/bash
# round 5, call 5: one-launch GroupNorm for HW <= 256 (parity + interleaved A/B against the previous library), the f16mx routing threshold with the tall tile
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/r05_call5.log; : > $LOG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_planes.py tests/test_gpu_backward.py tests/test_gpu_f16mx_model.py -m gpu -q -p no:cacheprovider --durations=5 2>&1 | tail -9 | tee -a $LOG
timeout 900 python -m pytest tests/test_gpu_model.py -k "tiny or sd21_shaped or 50_steps or plane_handover" -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $LOG
TAGS="prev" ENVS="DDPO_X=0;DDPO_MX_MIN_K=1280" ROUNDS=3 LOG=r05_ab_gn_fused.log bash tools/ab_bench.sh 2>&1 | tail -14 | tee -a $LOG
