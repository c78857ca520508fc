#!/bin/bash
# round-2 closing run: whole GPU suite, smoke, headline bench, kernel stats of sampling + train, full-scale entrypoint epochs
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/r02_pytest_gpu_full.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest_gpu_full.log; tail -6 gpurun_out/r02_pytest_gpu_full.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/r02_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r02_smoke.log; tail -3 gpurun_out/r02_smoke.log
timeout 600 python bench.py > gpurun_out/r02_bench_final.log 2>&1; echo "exit $?" >> gpurun_out/r02_bench_final.log; tail -2 gpurun_out/r02_bench_final.log | cut -c1-600
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s2 -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train-extra > $R/gpurun_out/prof_s2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t2 -o bench -- python $R/bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_t2.log 2>&1
cd $R
for d in prof_s2 prof_t2; do f=$(find gpurun_out/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r02_final_${d}_kernel_stats.md "round 2 final: $d" && find gpurun_out/$d -name "*.db" -delete; done
head -14 gpurun_out/r02_final_prof_s2_kernel_stats.md | cut -c1-160
( cd /tmp && rm -rf e2e_r02 && mkdir e2e_r02 && cd e2e_r02 && DDPO_ALLOW_SYNTHETIC=1 timeout 400 python $R/pipeline/policy_gradient.py --dataset compressed-animals --num_train_epochs 2 --save_freq 1000 --logbase /tmp/e2e_r02/run > $R/gpurun_out/r02_e2e_entrypoint_full_scale.log 2>&1 ); grep -E "sample \]|train steps|mean reward" gpurun_out/r02_e2e_entrypoint_full_scale.log | tail -8
