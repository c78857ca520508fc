#!/bin/bash
# validation of the row-loader weight gradient as the default: every test that runs a backward pass + the train bench with and without it
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_backward.py tests/test_gpu_planes.py tests/test_gpu_train_parity.py tests/test_fused_micro_steps.py tests/test_gpu_entrypoint.py tests/test_gpu_learning.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r02_pytest_wgrad_rows.log 2>&1; tail -3 gpurun_out/r02_pytest_wgrad_rows.log | cut -c1-200
for r in 0 1; do
  DDPO_WGRAD_ROWS=$r timeout 200 python bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train WGRAD_ROWS=$r', d['value'], d['ms_per_step'])"
done | tee gpurun_out/r02_ab_wgrad_rows_train.log
