#!/bin/bash
# round-4 second GPU call: attention probe, the tests that failed / are new, sampling A/Bs, kernel stats of the new default.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd tools/native && timeout 120 ./kernel_probe attn 16 10 > ../../gpurun_out/r04_probe_attn2.log 2>&1; tail -9 ../../gpurun_out/r04_probe_attn2.log)
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py tests/test_gpu_aesthetic.py tests/test_gpu_f16mx_model.py tests/test_gpu_headline_geometry.py \
  tests/test_gpu_model.py tests/test_gpu_rwr.py tests/test_gpu_train_parity.py tests/test_gpu_backward.py -m gpu -q --maxfail=40 -p no:cacheprovider --durations=10 \
  > gpurun_out/r04_pytest_gpu_call2.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_pytest_gpu_call2.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" gpurun_out/r04_pytest_gpu_call2.log | cut -c1-220 | tail -40
grep -E "^\[" gpurun_out/r04_pytest_gpu_call2.log | grep -E "headline|train parity|rwr sd15|sd21 96x96|f16mx" | cut -c1-330
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for i in 1 2; do
  for cfg in "new:" "copies:DDPO_SKIP_INPLACE=0" "bf16x3:DDPO_DATAPATH=bf16x3"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout 400 $B > gpurun_out/r04_ab2_$name.log 2>&1
    line=$(grep '^{"metric"' gpurun_out/r04_ab2_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "AB sample $i $name: $line" | tee -a gpurun_out/r04_ab_call2.log
    [ -z "$line" ] || true; tail -3 gpurun_out/r04_ab2_$name.log | cut -c1-300 >> gpurun_out/r04_ab_call2_tails.log
  done
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s4 -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train-extra --no-alt-datapath-extra > $R/gpurun_out/prof_s4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t4 -o bench -- python $R/bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_t4.log 2>&1
cd $R
for d in prof_s4 prof_t4; do f=$(find gpurun_out/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/r04_mid_${d}_kernel_stats.md "round 4 mid: $d" && find gpurun_out/$d -name "*.db" -delete; done
head -30 gpurun_out/r04_mid_prof_s4_kernel_stats.md | cut -c1-170
head -30 gpurun_out/r04_mid_prof_t4_kernel_stats.md | cut -c1-170
