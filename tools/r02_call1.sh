#!/bin/bash
# round-2 baseline: headline bench, train bench, rocprof kernel stats of both, the round-2 tests
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r02_bench_base.log 2>&1; echo "exit $?" >> gpurun_out/r02_bench_base.log; tail -2 gpurun_out/r02_bench_base.log | cut -c1-1500
timeout 300 python bench.py --mode train --no-cpu-baseline > gpurun_out/r02_bench_train_base.log 2>&1; tail -1 gpurun_out/r02_bench_train_base.log | cut -c1-800
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train-extra > $R/gpurun_out/prof_s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t -o bench -- python $R/bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_t.log 2>&1
cd $R
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out -name "*.db" -size +20M -delete
for d in prof_s prof_t; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; [ -n "$f" ] && head -25 "$f" | cut -c1-200; done
timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_aesthetic.py tests/test_gpu_planes.py -m gpu -q -p no:cacheprovider > gpurun_out/r02_pytest_new.log 2>&1; tail -5 gpurun_out/r02_pytest_new.log
