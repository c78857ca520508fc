#!/bin/bash
# round-3 closing run: whole GPU suite, smoke, headline bench (with hbm_kernels / stamped traffic), kernel stats of sampling + train,
# PMC traffic of the U-Net GEMM family, C5 lines on both datapaths, epoch line.   gpurun --timeout 3000 -- 'bash tools/r03_final.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --durations=12 > gpurun_out/r03_pytest_gpu_full.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03_pytest_gpu_full.log; tail -22 gpurun_out/r03_pytest_gpu_full.log | cut -c1-200
fi
timeout 300 python __graft_entry__.py smoke > gpurun_out/r03_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r03_smoke.log; tail -3 gpurun_out/r03_smoke.log
# PMC traffic first: bench.py quotes it only when it was taken on the current kernel sources
N=2 timeout 900 bash tools/pmc_unet_traffic.sh > gpurun_out/r03_pmc_traffic.log 2>&1; tail -12 gpurun_out/r03_pmc_traffic.log | cut -c1-200
timeout 700 python bench.py > gpurun_out/r03_bench_final.log 2>&1; echo "exit $?" >> gpurun_out/r03_bench_final.log; tail -2 gpurun_out/r03_bench_final.log | cut -c1-900
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s3 -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train-extra > $R/gpurun_out/prof_s3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t3 -o bench -- python $R/bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_t3.log 2>&1
cd $R
for d in prof_s3 prof_t3; do f=$(find gpurun_out/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r03_final_${d}_kernel_stats.md "round 3 final: $d" && find gpurun_out/$d -name "*.db" -delete; done
head -16 gpurun_out/r03_final_prof_s3_kernel_stats.md | cut -c1-170
timeout 400 python bench.py --mode train --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r03_bench_train_final.log 2>&1; tail -1 gpurun_out/r03_bench_train_final.log | cut -c1-400
timeout 400 python bench.py --mode epoch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r03_bench_epoch.log 2>&1; tail -1 gpurun_out/r03_bench_epoch.log | cut -c1-400
timeout 500 bash tools/pmc_wgrad.sh > gpurun_out/r03_pmc_wgrad.log 2>&1; tail -8 gpurun_out/r03_pmc_wgrad.log | cut -c1-300
(cd tools/native && for m in gemm2 mx wgrad; do PROBE_WKBLK=1 timeout 200 ./kernel_probe $m 16 10 > ../../gpurun_out/r03_final_probe_$m.log 2>&1; tail -1 ../../gpurun_out/r03_final_probe_$m.log; done)
# BASELINE configs[4] (C5): SD-2.1 768^2 on the fp32-class datapath and on the config's named dtype (bfloat16 -> single-pass bf16 MFMA)
timeout 500 python bench.py --model sd21 --resolution 768 --no-cpu-baseline --no-train-extra > gpurun_out/r03_bench_c5_sd21_768_bf16x3.log 2>&1; tail -1 gpurun_out/r03_bench_c5_sd21_768_bf16x3.log | cut -c1-300
timeout 500 python bench.py --model sd21 --resolution 768 --datapath bf16 --no-cpu-baseline --no-train-extra > gpurun_out/r03_bench_c5_sd21_768_bf16.log 2>&1; tail -1 gpurun_out/r03_bench_c5_sd21_768_bf16.log | cut -c1-300
