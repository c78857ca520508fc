#!/bin/bash
# round 3, GPU call: wide wgrad tile — backward / wgrad tests, train bench A/B (DDPO_WGRAD_WIDE=0/1 interleaved)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_planes.py tests/test_gpu_train_parity.py -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_train_parity.py::test_train_step_sd21_full_size_bf16x3 > gpurun_out/r03_pytest_wgrad.log 2>&1; tail -4 gpurun_out/r03_pytest_wgrad.log | cut -c1-300
for v in 0 1 0 1; do
  DDPO_WGRAD_WIDE=$v timeout 400 python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train WGRAD_WIDE=$v', d['value'], d['ms_per_step'])"
done | tee gpurun_out/r03_ab_wgrad_wide.log
