#!/bin/bash
P=tools/native/kernel_probe
for cold in 1 0; do for ad in 0 160; do
  echo "== ADEEP=$ad cold=$cold"
  for c in d0 d2 c10 d4 d6; do DDPO_APL_ADEEP=$ad PROBE_COLD=$cold PROBE_ONLY=$c timeout 120 $P gemm2 16 20 2>&1 | grep -E "^gemm|^conv"; done
done; done 2>&1 | cut -c1-150 | tee gpurun_out/r02_probe_adeep.log
