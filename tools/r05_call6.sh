#!/bin/bash
# round 5, call 6: TIMING-ONLY experiment (private library, wrong results by construction): channel-block-major order of the activation stream of the
# 3x3 convolutions (all nine taps of a 32-channel block back to back, so that the tap re-reads hit the XCD's L2 instead of the Infinity Cache);
# fresh per-launch timeline; train bench of the current tree
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/r05_call6.log; : > $LOG
L=ddpo_amd/libddpo_hip.so; cp $L /tmp/new.so
for v in new cimajor new cimajor; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp tools/native/libddpo_hip_$v.so $L; fi
  echo "== lib=$v" | tee -a $LOG
  (cd tools/native && timeout 200 ./kernel_probe mx 16 5 2>&1 | grep "^conv 3x3" | cut -c1-140) | tee -a $LOG
done
cp /tmp/new.so $L
bash tools/timeline.sh 2>&1 | tail -3 | tee -a $LOG
timeout 600 python bench.py --mode train --steps 6 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep '^{"metric"' | tee gpurun_out/r05_bench_train_mid.log | cut -c1-300 | tee -a $LOG
