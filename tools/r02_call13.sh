#!/bin/bash
P=tools/native/kernel_probe
for cfg in "DDPO_GEMM_WIDE=1" "DDPO_GEMM_WIDE=0"; do
  echo "== $cfg"
  for c in d0 d2 c10; do env $cfg PROBE_COLD=1 PROBE_ONLY=$c timeout 120 $P gemm2 16 20 2>&1 | grep -E "^gemm|^conv"; done
done 2>&1 | cut -c1-150 | tee gpurun_out/r02_probe_k320_tiles.log
