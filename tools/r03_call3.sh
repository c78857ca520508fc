#!/bin/bash
# round 3, GPU call: k-blocked weight planes (w_layout = 1): bit-identity tests, probe A/B, bench A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
P=tools/native/kernel_probe
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_planes.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r03_pytest_kblk.log 2>&1; tail -4 gpurun_out/r03_pytest_kblk.log | cut -c1-300
{
for cold in 0 1; do for rep in 1 2; do for kb in 0 1; do
  echo "== rep=$rep cold=$cold WKBLK=$kb"; PROBE_COLD=$cold PROBE_WKBLK=$kb timeout 120 $P gemm2 16 10 | grep -v "^#"
done; done; done
} > gpurun_out/r03_probe_wkblk.log 2>&1
grep -c "bit-identical" gpurun_out/r03_probe_wkblk.log; grep -c FAIL gpurun_out/r03_probe_wkblk.log
for v in 0 1 0 1; do
  DDPO_W_KBLOCKED=$v timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample W_KBLOCKED=$v', d['value'], d['ms_per_step'])"
done | tee gpurun_out/r03_ab_wkblk_bench.log
for v in 0 1; do
  DDPO_W_KBLOCKED=$v timeout 300 python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train W_KBLOCKED=$v', d['value'], d['ms_per_step'])"
done | tee -a gpurun_out/r03_ab_wkblk_bench.log
