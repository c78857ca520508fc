#!/bin/bash
# round 5, call 2: oracle thread sweep, tall GEGLU tile (parity + interleaved A/B), the new parity tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/r05_call2.log; : > $LOG
timeout 300 python tools/oracle_threads.py 16 32 64 128 2>&1 | tee -a $LOG
export DDPO_PARITY_LOG=$PWD/gpurun_out/r05_parity_call2.log; : > $DDPO_PARITY_LOG
timeout 900 python -m pytest tests/test_gpu_bf16.py -k "geglu" tests/test_gpu_f16mx.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider --durations=8 2>&1 | tail -16 | tee -a $LOG
timeout 900 python -m pytest tests/test_gpu_train_parity.py -k "sd15_full_size" tests/test_gpu_headline_geometry.py -m gpu -q -x -p no:cacheprovider --durations=8 2>&1 | tail -12 | tee -a $LOG
ENVS="DDPO_GEGLU_TALL=0;DDPO_GEGLU_TALL=1" ROUNDS=2 LOG=r05_ab_geglu_tall.log bash tools/ab_bench.sh 2>&1 | tail -6 | tee -a $LOG
