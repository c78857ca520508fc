#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_planes.py tests/test_gpu_train_parity.py -m gpu -q -p no:cacheprovider -k "wgrad or grads or train" > gpurun_out/r02_pytest_c8.log 2>&1; tail -4 gpurun_out/r02_pytest_c8.log | cut -c1-300
for rep in 1 2; do for dp in 0 1; do
  DDPO_WGRAD_DEEP=$dp timeout 300 python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train WGRAD_DEEP=$dp', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r02_ab_wgrad_deep.log
timeout 600 python bench.py --model sd21 --resolution 768 --steps 1 --warmup 1 --no-cpu-baseline --no-train-extra > gpurun_out/r02_bench_c5_sd21_768.log 2>&1; tail -1 gpurun_out/r02_bench_c5_sd21_768.log | cut -c1-700
