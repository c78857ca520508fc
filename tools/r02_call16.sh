#!/bin/bash
mkdir -p gpurun_out
for c in c0 c1 d0; do
  CASE=$c MODES="7" bash tools/pmc_probe.sh > gpurun_out/r02_pmc_probe_$c.log 2>&1
  grep -E "^mode" gpurun_out/r02_pmc_probe_$c.log | cut -c1-700
done
