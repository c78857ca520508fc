#!/bin/bash
# In-model A/B of the nontemporal-output experiment (tools/native `make exp`): the sampling bench with the shipped library vs the private
# builds whose GEMM output stage writes fp32 outputs (nto) / fp32 + plane outputs (nto2) with nontemporal stores.  Interleaved on ONE box;
# the library file is swapped in the box's scratch copy only.
mkdir -p gpurun_out; export TMPDIR=/tmp
L=ddpo_amd/libddpo_hip.so
cp $L /tmp/base.so
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for round in 1 2; do
  for v in base nto nto2; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp tools/native/libddpo_hip_$v.so $L; fi
    line=$(timeout 400 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "sample $v (round $round): $line" | tee -a gpurun_out/r04_ab_nt.log
  done
done
cp /tmp/base.so $L
