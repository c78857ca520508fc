#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
P=tools/native/kernel_probe
for cfg in "default" "DDPO_GEMM_WIDE=0" "DDPO_GEMM_WIDE=0 DDPO_GEMM_BIG_MIN=100000"; do
  echo "== $cfg"
  for c in d11 d12 d7 d6; do
    if [ "$cfg" = "default" ]; then PROBE_COLD=1 PROBE_ONLY=$c timeout 120 $P gemm2 16 20 2>&1 | grep -E "^gemm|^conv"
    else env $cfg PROBE_COLD=1 PROBE_ONLY=$c timeout 120 $P gemm2 16 20 2>&1 | grep -E "^gemm|^conv"; fi
  done
done > gpurun_out/r02_probe_tiles_small.log 2>&1
cut -c1-140 gpurun_out/r02_probe_tiles_small.log
timeout 600 python -m pytest tests/test_gpu_learning.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r02_pytest_learning.log 2>&1; tail -4 gpurun_out/r02_pytest_learning.log | cut -c1-600
N=2 bash tools/pmc_unet_traffic.sh > gpurun_out/r02_pmc_unet_traffic.log 2>&1; tail -25 gpurun_out/r02_pmc_unet_traffic.log
