#!/bin/bash
# round 3, GPU call 2: phase timing + ablations of the plane-fed k-loops (instrumented private build of the library)
mkdir -p gpurun_out; export TMPDIR=/tmp
P=tools/native/kernel_probe_timing
{
for cold in 0 1; do
  for abl in 0 1 2 3; do
    echo "== cold=$cold TALL_ROT=0 ABL=$abl"; PROBE_COLD=$cold DDPO_TALL_ROT=0 DDPO_DBG_ABL=$abl timeout 120 $P ktime 16 | grep -v "^#"
  done
  echo "== cold=$cold TALL_ROT=1 ABL=0"; PROBE_COLD=$cold DDPO_TALL_ROT=1 DDPO_DBG_ABL=0 timeout 120 $P ktime 16 | grep -v "^#"
done
} > gpurun_out/r03_ktime.log 2>&1
cat gpurun_out/r03_ktime.log | cut -c1-330
