#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fused_micro_steps.py tests/test_gpu_entrypoint.py tests/test_gpu_learning.py -m gpu -q -p no:cacheprovider > gpurun_out/r02_pytest_c15.log 2>&1; tail -4 gpurun_out/r02_pytest_c15.log | cut -c1-300
timeout 300 python bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_train_final.log 2>&1; tail -1 gpurun_out/r02_bench_train_final.log | cut -c1-500
timeout 600 python bench.py --mode epoch --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-900 | tee gpurun_out/r02_bench_epoch.log
( cd /tmp && rm -rf e2e_r02 && mkdir e2e_r02 && cd e2e_r02 && DDPO_ALLOW_SYNTHETIC=1 timeout 400 python $R/pipeline/policy_gradient.py --dataset compressed-animals --num_train_epochs 3 --save_freq 1000 --logbase /tmp/e2e_r02/run > $R/gpurun_out/r02_e2e_entrypoint_full_scale.log 2>&1 ); grep -E "sample \]|train steps|mean reward" gpurun_out/r02_e2e_entrypoint_full_scale.log | tail -9
