#!/bin/bash
# Third closing run of round 4 (library with the two-phase output stage, ABI v12): PMC traffic of the GEMM family on the current sources (stamped,
# so the bench line carries it), the default bench, kernel stats of the same command, the train line.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
N=2 timeout 420 bash tools/pmc_unet_traffic.sh > gpurun_out/r04_pmc_traffic_close3.log 2>&1; tail -6 gpurun_out/r04_pmc_traffic_close3.log | cut -c1-160
python tools/stamp_traffic.py f16mx > /dev/null 2>&1 && cp profiles/roofline_traffic.json gpurun_out/roofline_traffic_close3.json
timeout 600 python bench.py > gpurun_out/r04_bench_close3.log 2>&1; echo "exit $?" >> gpurun_out/r04_bench_close3.log; tail -2 gpurun_out/r04_bench_close3.log | cut -c1-1800
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s6 -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train-extra --no-alt-datapath-extra > $R/gpurun_out/prof_s6.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/prof_s6 -name "*.db" | head -1) gpurun_out/r04_close3_sampling_kernel_stats.md "round 4, second closing run: sampling, shipped datapath (f16mx), two-phase GEMM output stage; bench.py --steps 1 --warmup 0 (includes the graph capture's warm-up forwards)"
python tools/rocpd_timeline.py $(find gpurun_out/prof_s6 -name "*.db" | head -1) gpurun_out/r04_close3_timeline_sampling_step.txt
find gpurun_out/prof_s6 -name "*.db" -delete
timeout 300 python bench.py --mode train --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r04_bench_train_close3.log 2>&1; tail -1 gpurun_out/r04_bench_train_close3.log | cut -c1-400
