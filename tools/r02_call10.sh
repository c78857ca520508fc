#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rccl_single_rank.py -m gpu -q -p no:cacheprovider > gpurun_out/r02_pytest_rccl.log 2>&1; tail -15 gpurun_out/r02_pytest_rccl.log | cut -c1-400
for tm in 2 1; do
DDPO_APL_TALL=$tm timeout 600 python bench.py --model sd21 --resolution 768 --steps 1 --warmup 1 --no-cpu-baseline --no-train-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd21 TALL=$tm', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done | tee gpurun_out/r02_ab_sd21_tall_rule.log
