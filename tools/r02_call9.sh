#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_planes.py tests/test_gpu_backward.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > gpurun_out/r02_pytest_c9.log 2>&1; tail -4 gpurun_out/r02_pytest_c9.log | cut -c1-300
timeout 300 python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'])"
