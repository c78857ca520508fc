#!/bin/bash
# round 3, GPU call: f16 + MX-fp8 MFMA-mix ablation of the plane-fed k-loops (timing builds), and the tests still owed a GPU run
mkdir -p gpurun_out; export TMPDIR=/tmp
cd tools/native
{ echo "== bf16x3 (shipping mix)"; timeout 200 ./kernel_probe_timing ktime 16; echo "== f16 + MX-fp8 mix ablation (wrong results)"; timeout 200 ./kernel_probe_timing_mx ktime 16; } > ../../gpurun_out/r03_ktime_mx_ablation.log 2>&1
cd ../..
grep -c ktime gpurun_out/r03_ktime_mx_ablation.log
timeout 900 python -m pytest tests/test_rwr_pipeline.py tests/test_gpu_entrypoint.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_rwr2.log 2>&1; tail -5 gpurun_out/r03_pytest_rwr2.log | cut -c1-300
