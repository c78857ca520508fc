#!/bin/bash
# round 3, GPU call: bias gradient folded into the weight-gradient launch + GEGLU fused on the training forward — tests, then train A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1100 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_planes.py tests/test_fused_micro_steps.py tests/test_gpu_train_parity.py tests/test_gpu_entrypoint.py -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_train_parity.py::test_train_step_sd21_full_size_bf16x3 > gpurun_out/r03_pytest_fuse.log 2>&1; tail -4 gpurun_out/r03_pytest_fuse.log | cut -c1-300
for v in 0 1 0 1; do
  DDPO_AB_FUSE_BIAS=$v DDPO_AB_TRAIN_GEGLU=$v timeout 400 python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train FUSE_BIAS+TRAIN_GEGLU=$v', d['value'], d['ms_per_step'])"
done | tee gpurun_out/r03_ab_train_fuse_bias_geglu.log
