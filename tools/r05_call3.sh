#!/bin/bash
# round 5, call 3: f16mx tall tile (probe: bit-identity + timing per layer; parity tests; interleaved bench A/B) + the tests call 2 deselected by mistake
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/r05_call3.log; : > $LOG
(cd tools/native && timeout 300 ./kernel_probe mx 16 5) 2>&1 | tee -a $LOG
export DDPO_PARITY_LOG=$PWD/gpurun_out/r05_parity_call3.log; : > $DDPO_PARITY_LOG
timeout 900 python -m pytest tests/test_gpu_f16mx.py tests/test_gpu_f16mx_model.py tests/test_gpu_planes.py -m gpu -q -x -p no:cacheprovider --durations=5 2>&1 | tail -10 | tee -a $LOG
ENVS="DDPO_MX_TALL=0;DDPO_MX_TALL=1;DDPO_MX_TALL=2" ROUNDS=2 LOG=r05_ab_mx_tall.log bash tools/ab_bench.sh 2>&1 | tail -8 | tee -a $LOG
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_headline_geometry.py -m gpu -q -x -p no:cacheprovider --durations=8 2>&1 | tail -14 | tee -a $LOG
