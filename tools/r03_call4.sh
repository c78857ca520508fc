#!/bin/bash
# round 3, GPU call: k-blocked ACTIVATION planes: tests, bench A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_planes.py tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_fused_micro_steps.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_model.py::test_sampler_sd21_full_size_96x96_graph_path > gpurun_out/r03_pytest_akblk.log 2>&1; tail -4 gpurun_out/r03_pytest_akblk.log | cut -c1-300
for v in 0 1 0 1; do
  DDPO_A_KBLOCKED=$v timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample A_KBLOCKED=$v', d['value'], d['ms_per_step'])"
done | tee gpurun_out/r03_ab_akblk_bench.log
for v in 0 1 0 1; do
  DDPO_A_KBLOCKED=$v timeout 300 python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train A_KBLOCKED=$v', d['value'], d['ms_per_step'])"
done | tee -a gpurun_out/r03_ab_akblk_bench.log
