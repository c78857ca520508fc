#!/bin/bash
# Plane hand-over behind the attention / FF2 (ABI v12): operator + model tests, then the sampling bench A/B interleaved on one box.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "plane_emitting or attention_bf16x3 or plane_handover or skip_concat or context_kv" 2>&1 | tail -6
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for round in 1 2; do
  for v in "0 0" "0 1" "1 0" "1 1"; do
    set -- $v
    line=$(DDPO_ATTN_PLANES=$1 DDPO_H3_PLANES=$2 timeout 400 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "sample ATTN_PLANES=$1 H3_PLANES=$2 (round $round): $line" | tee -a gpurun_out/r04_ab_handover.log
  done
done
