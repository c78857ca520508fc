#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_learning.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r02_pytest_planes_learning.log 2>&1; tail -15 gpurun_out/r02_pytest_planes_learning.log
timeout 400 python bench.py --no-cpu-baseline --no-train-extra > gpurun_out/r02_bench_planes_out.log 2>&1; tail -1 gpurun_out/r02_bench_planes_out.log | cut -c1-400
timeout 300 python tools/unet_gemm_breakdown.py 16 --ab > gpurun_out/r02_gemm_breakdown_ab2.log 2>&1; head -3 gpurun_out/r02_gemm_breakdown_ab2.log
