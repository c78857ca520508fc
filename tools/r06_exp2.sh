#!/bin/bash
# round 6, call 2: single-pass bf16 plane-fed kernels — probe (bit identity + timing vs the fp32-fed single-pass kernel), pytest, C5 bench A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_exp2.log
{
echo "== kernel_probe x1 (batch 16)"; timeout 600 tools/native/kernel_probe x1 16 5
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_bf16_planes.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
for r in 1 2; do
for e in 0 1; do
  echo "== C5 bench DDPO_BF16_PLANES=$e (round $r)"
  DDPO_BF16_PLANES=$e timeout 900 python bench.py --model sd21 --resolution 768 --datapath bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-train-extra --no-alt-datapath-extra 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print(d['value'], d['unit'], d['ms_per_step'], 'ms | gemm family', r.get('achieved'), 'TF frac', r.get('frac'), '| dtype', d['dtype'])"
done; done
} > $L 2>&1
tail -12 $L
