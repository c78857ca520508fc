#!/bin/bash
# round 3, GPU call: RWR kernels / train step parity against the oracle (tiny + full-size SD-1.5)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rwr.py -m gpu -q -p no:cacheprovider -s --durations=5 > gpurun_out/r03_pytest_gpu_rwr.log 2>&1; tail -25 gpurun_out/r03_pytest_gpu_rwr.log | cut -c1-300
