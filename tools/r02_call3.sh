#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
P=tools/native/kernel_probe
for cold in 0 1; do
for n in 0 40; do
  echo "== N160=$n cold=$cold"
  for c in d0 d2 d11 d12 d13 c10; do
    PROBE_COLD=$cold DDPO_GEMM_N160=$n PROBE_ONLY=$c timeout 120 $P gemm2 16 20 2>&1 | grep -E "gemm|conv"
  done
done
done > gpurun_out/r02_probe_n160.log 2>&1
cat gpurun_out/r02_probe_n160.log
DDPO_ALLOW_SYNTHETIC=1 timeout 500 python tools/learning_sweep.py > gpurun_out/r02_learning_sweep.md 2> gpurun_out/r02_learning_sweep.err; cat gpurun_out/r02_learning_sweep.md; tail -3 gpurun_out/r02_learning_sweep.err
