#!/bin/bash
# round-4 first GPU call: attention probe, whole GPU suite on the new defaults, interleaved A/Bs of the host-level switches.
#   gpurun --timeout 2400 -- 'bash tools/r04_call1.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd tools/native && timeout 120 ./kernel_probe attn 16 10 > ../../gpurun_out/r04_probe_attn.log 2>&1; tail -9 ../../gpurun_out/r04_probe_attn.log)
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=15 > gpurun_out/r04_pytest_gpu_call1.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_pytest_gpu_call1.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" gpurun_out/r04_pytest_gpu_call1.log | cut -c1-220 | tail -50
grep -E "^\[|\] " gpurun_out/r04_pytest_gpu_call1.log | grep -E "headline|train parity|rwr sd15|sd21 96x96" | cut -c1-330
fi
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for i in 1 2; do
  for cfg in "new:" "copies:DDPO_SKIP_INPLACE=0" "bf16x3:DDPO_DATAPATH=bf16x3"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    line=$(env $envs timeout 400 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1)
    echo "AB sample $i $name: $line" | tee -a gpurun_out/r04_ab_call1.log
  done
done
T="python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-roofline"
for i in 1 2; do
  for cfg in "new:" "dgrad_old:DDPO_DGRAD_FWD=0" "bf16x3:DDPO_DATAPATH=bf16x3"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    line=$(env $envs timeout 400 $T 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1)
    echo "AB train $i $name: $line" | tee -a gpurun_out/r04_ab_call1.log
  done
done
