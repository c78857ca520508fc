#!/bin/bash
# round-6 closing run: whole GPU suite (parity margins recorded), smoke, PMC traffic of the shipped datapath, headline bench, kernel stats of
# sampling + train, train / epoch lines (epoch also under the forced one-rank RCCL group: GradBucketer path), C5 lines, probes.
#   gpurun --timeout 3000 -- 'bash tools/r06_final.sh'          SKIP_TESTS=1 skips the suite; ONLY_TESTS=1 stops after suite + smoke
# The PMC traffic pass stamps profiles/roofline_traffic.json with the hashes of csrc/: it must be the LAST thing that happens to the kernel sources
# of the round — commit the stamped json and nothing under ddpo_amd/csrc after it.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
rm -f gpurun_out/r06_parity_margins.log
DDPO_PARITY_LOG=$R/gpurun_out/r06_parity_margins.log timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --durations=15 > gpurun_out/r06_pytest_gpu_full.log 2>&1
echo "pytest exit $?" >> gpurun_out/r06_pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" gpurun_out/r06_pytest_gpu_full.log | cut -c1-220 | tail -30
cat gpurun_out/r06_parity_margins.log | cut -c1-300
fi
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r06_smoke.log; tail -3 gpurun_out/r06_smoke.log
[ "${ONLY_TESTS:-0}" = "1" ] && exit 0
# PMC traffic first: bench.py quotes it only when it was taken on the current kernel sources
N=2 timeout 900 bash tools/pmc_unet_traffic.sh > gpurun_out/r06_pmc_traffic.log 2>&1; tail -12 gpurun_out/r06_pmc_traffic.log | cut -c1-200
# stamp it into profiles/roofline_traffic.json ON THE BOX so the bench line below carries it (re-run tools/stamp_traffic.py locally afterwards: adds the git commit)
python tools/stamp_traffic.py f16mx > /dev/null 2>&1 && cp profiles/roofline_traffic.json gpurun_out/roofline_traffic_stamped.json
timeout 900 python bench.py > gpurun_out/r06_bench_final.log 2>&1; echo "exit $?" >> gpurun_out/r06_bench_final.log; tail -2 gpurun_out/r06_bench_final.log | cut -c1-1500
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s6 -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train-extra --no-alt-datapath-extra > $R/gpurun_out/prof_s6.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t6 -o bench -- python $R/bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_t6.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/prof_s6 -name "*.db" | head -1) gpurun_out/r06_final_sampling_kernel_stats.md "round 6 final: sampling, shipped datapath (f16mx), bench.py --steps 1 --warmup 0 (includes the graph capture's warm-up forwards)"
python tools/rocpd_summary.py $(find gpurun_out/prof_t6 -name "*.db" | head -1) gpurun_out/r06_final_train_kernel_stats.md "round 6 final: train, shipped datapath (f16mx), bench.py --mode train --steps 1 --warmup 0"
python tools/rocpd_timeline.py $(find gpurun_out/prof_s6 -name "*.db" | head -1) gpurun_out/r06_final_timeline_sampling_step.txt      # per-launch view of the last step
find gpurun_out/prof_s6 gpurun_out/prof_t6 -name "*.db" -delete
head -24 gpurun_out/r06_final_sampling_kernel_stats.md | cut -c1-170
timeout 400 python bench.py --mode train --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_train_final.log 2>&1; tail -1 gpurun_out/r06_bench_train_final.log | cut -c1-400
timeout 400 python bench.py --mode epoch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_epoch.log 2>&1; tail -1 gpurun_out/r06_bench_epoch.log | cut -c1-400
# the same epoch with the gradient all-reduce on RCCL: forced one-rank nccl group -> the bucketed all-reduce behind the backward (GradBucketer), then the blocking one
DDPO_FORCE_DIST=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --mode epoch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_epoch_rccl1_bucketed.log 2>&1; tail -1 gpurun_out/r06_bench_epoch_rccl1_bucketed.log | cut -c1-400
DDPO_FORCE_DIST=1 DDPO_GRAD_OVERLAP=0 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --mode epoch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_epoch_rccl1_blocking.log 2>&1; tail -1 gpurun_out/r06_bench_epoch_rccl1_blocking.log | cut -c1-400
(cd tools/native && timeout 120 ./kernel_probe attn 16 10 > ../../gpurun_out/r06_final_probe_attn.log 2>&1; tail -3 ../../gpurun_out/r06_final_probe_attn.log; for m in gemm2 mx x1; do PROBE_WKBLK=1 timeout 300 ./kernel_probe $m 16 10 > ../../gpurun_out/r06_final_probe_$m.log 2>&1; tail -1 ../../gpurun_out/r06_final_probe_$m.log; done)
# BASELINE configs[4] (C5): SD-2.1 768^2 on the shipped datapath and on the config's named dtype (bfloat16 -> single-pass bf16 MFMA)
timeout 500 python bench.py --model sd21 --resolution 768 --no-cpu-baseline --no-train-extra --no-alt-datapath-extra > gpurun_out/r06_bench_c5_sd21_768_f16mx.log 2>&1; tail -1 gpurun_out/r06_bench_c5_sd21_768_f16mx.log | cut -c1-300
timeout 500 python bench.py --model sd21 --resolution 768 --datapath bf16 --no-cpu-baseline --no-train-extra --no-alt-datapath-extra > gpurun_out/r06_bench_c5_sd21_768_bf16.log 2>&1; tail -1 gpurun_out/r06_bench_c5_sd21_768_bf16.log | cut -c1-300
timeout 300 python tools/unet_gemm_breakdown.py 16 --ab > gpurun_out/r06_gemm_breakdown_ab.log 2>&1; head -30 gpurun_out/r06_gemm_breakdown_ab.log | cut -c1-200
timeout 300 bash tools/pmc_unet_clock.sh > /dev/null 2>&1; cp gpurun_out/pmc_unet_clock.md gpurun_out/r06_final_pmc_unet_clock.md; head -12 gpurun_out/pmc_unet_clock.md | cut -c1-200
timeout 300 bash tools/pmc_unet_l2_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_unet_l2_mfma.md gpurun_out/r06_final_pmc_unet_l2_mfma.md
