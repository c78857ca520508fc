#!/bin/bash
# round-4 third GPU call: both attention variants (probe + tests), gradient parity at size without f16mx data gradients, train / sample A/B by datapath.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd tools/native && timeout 120 ./kernel_probe attn 16 10 > ../../gpurun_out/r04_probe_attn3.log 2>&1; tail -14 ../../gpurun_out/r04_probe_attn3.log)
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_model.py tests/test_gpu_rwr.py tests/test_gpu_train_parity.py tests/test_gpu_f16mx_model.py tests/test_gpu_backward.py \
  tests/test_fused_micro_steps.py -m gpu -q --maxfail=40 -p no:cacheprovider --durations=8 -k "not sd21_full_size_96x96 and not vae_sd_decode" \
  > gpurun_out/r04_pytest_gpu_call3.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_pytest_gpu_call3.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" gpurun_out/r04_pytest_gpu_call3.log | cut -c1-220 | tail -40
grep -E "^\[" gpurun_out/r04_pytest_gpu_call3.log | grep -E "train parity|rwr sd15|attention bwd|attention fwd" | cut -c1-330 | head -60
T="python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-roofline"
for i in 1 2; do
  for cfg in "f16mx:" "bf16x3:DDPO_DATAPATH=bf16x3"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    line=$(env $envs timeout 400 $T 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "AB train $i $name: $line" | tee -a gpurun_out/r04_ab_call3.log
  done
done
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for cfg in "f16mx:" "bf16x3:DDPO_DATAPATH=bf16x3"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  line=$(env $envs timeout 400 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
  echo "AB sample $name: $line" | tee -a gpurun_out/r04_ab_call3.log
done
