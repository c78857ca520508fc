#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for m in 7 6 3; do
  DDPO_APL_MODE=$m timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample APL_MODE=$m', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r02_ab_apl_mode.log
for rep in 1 2; do for k in 10 8 16; do
  timeout 300 python bench.py --mode train --train-fuse $k --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train fuse=$k', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r02_ab_train_fuse.log
timeout 600 python bench.py --mode epoch --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-900 | tee gpurun_out/r02_bench_epoch.log
