#!/bin/bash
# round-4 exploration with the GPU minutes left after the closing run: f16mx routing threshold, PPO micro-step fusion width.
mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 2560 1920 1280 640; do
  echo "== MX_MIN_K=$k" | tee -a gpurun_out/r04_explore.log
  DDPO_MX_MIN_K=$k timeout 300 python tools/unet_gemm_breakdown.py 16 --ab 2>&1 | grep -E "^planes  :|^fp32-fed:" | tee -a gpurun_out/r04_explore.log
done
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for k in 2560 1280; do
  line=$(DDPO_MX_MIN_K=$k timeout 400 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
  echo "sample MX_MIN_K=$k: $line" | tee -a gpurun_out/r04_explore.log
done
for f in 16 17 25; do
  line=$(DDPO_TRAIN_FUSE=$f timeout 500 python bench.py --mode epoch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" 2>&1 | tail -1)
  echo "epoch TRAIN_FUSE=$f: $line ms" | tee -a gpurun_out/r04_explore.log
done
