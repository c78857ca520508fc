#!/bin/bash
# round 5, call 7: coalesced one-launch GroupNorm (parity + A/B against the first form)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/r05_call7.log; : > $LOG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_planes.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $LOG
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_f16mx_model.py -k "tiny or sd21_shaped or 50_steps or plane_handover or f16mx" -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $LOG
TAGS="prev" ROUNDS=3 LOG=r05_ab_gn_fused2.log bash tools/ab_bench.sh 2>&1 | tail -7 | tee -a $LOG
