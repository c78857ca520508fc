#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python tools/learning_sweep.py > gpurun_out/r02_learning_sweep.md 2> gpurun_out/r02_learning_sweep.err; cat gpurun_out/r02_learning_sweep.md; tail -3 gpurun_out/r02_learning_sweep.err
timeout 300 python tools/unet_gemm_breakdown.py 16 --ab > gpurun_out/r02_gemm_breakdown_ab.log 2>&1; tail -40 gpurun_out/r02_gemm_breakdown_ab.log
timeout 400 python -m pytest tests/test_gpu_entrypoint.py -m gpu -q -p no:cacheprovider -k "two_ranks" > gpurun_out/r02_pytest_2rank.log 2>&1; tail -5 gpurun_out/r02_pytest_2rank.log
