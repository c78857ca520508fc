#!/bin/bash
# round 5, call 4: 16-lane DPP LayerNorm (C = 320): parity tests + interleaved A/B against the previous library; the f16mx / planes / bf16 suites on the new kernels
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/r05_call4.log; : > $LOG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_planes.py tests/test_gpu_f16mx.py tests/test_gpu_f16mx_model.py -m gpu -q -p no:cacheprovider --durations=5 2>&1 | tail -14 | tee -a $LOG
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee -a $LOG
TAGS="prev" ROUNDS=3 LOG=r05_ab_ln16.log bash tools/ab_bench.sh 2>&1 | tail -8 | tee -a $LOG
