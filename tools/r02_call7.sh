#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_model.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > gpurun_out/r02_pytest_c7.log 2>&1; tail -6 gpurun_out/r02_pytest_c7.log | cut -c1-300
for rep in 1 2; do
  DDPO_PLANES_ALL=1 DDPO_GEMM_MID64=0 DDPO_TEMB_CACHE=0 timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample OLD', d['value'], d['ms_per_step'])"
  DDPO_TEMB_CACHE=0 timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample RULES', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample RULES+TEMB', d['value'], d['ms_per_step'])"
done | tee gpurun_out/r02_ab_rules_temb.log
timeout 300 python tools/unet_gemm_breakdown.py 16 > gpurun_out/r02_gemm_breakdown_rules.log 2>&1; head -30 gpurun_out/r02_gemm_breakdown_rules.log
