#!/bin/bash
# round 3, GPU call: f16mx operator tests + regression of the plane-fed / bf16 suites after the kernel edits; probe (final, 128-row tiles only)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f16mx.py tests/test_gpu_planes.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_f16mx.log 2>&1; tail -15 gpurun_out/r03_pytest_f16mx.log | cut -c1-300
cd tools/native && timeout 300 ./kernel_probe mx 16 10 > ../../gpurun_out/r03_probe_mx.log 2>&1; tail -3 ../../gpurun_out/r03_probe_mx.log | cut -c1-200
