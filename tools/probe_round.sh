#!/bin/bash
# Torch-free kernel sweep on the GPU box (seconds per run): spot checks + timings of the hot kernels through the C ABI.
#   gpurun --timeout 300 -- 'bash tools/probe_round.sh'
# PROBE_KNOBS="DDPO_GEMM_WIDE=0 DDPO_GEMM_BIG_MIN=128" adds one extra gemm sweep per listed VAR=value setting.
# PROBE_APL_MODES="1 2 6 3 7" / PROBE_COLD_MODES="6 7": plane-fed vs fp32-fed per k-loop variant, caches warm / flushed.
# First call of round 2:  PROBE_APL_MODES="6 3 7 14" PROBE_COLD_MODES="6 7 14" bash tools/probe_round.sh
mkdir -p gpurun_out
P=tools/native/kernel_probe
[ -x $P ] || make -C tools/native > gpurun_out/probe_build.log 2>&1
timeout 60 $P ppo > gpurun_out/probe_ppo.log 2>&1; echo "exit $?" >> gpurun_out/probe_ppo.log; tail -3 gpurun_out/probe_ppo.log
timeout 120 $P gemm ${PROBE_BATCH:-16} ${PROBE_ITERS:-10} > gpurun_out/probe_gemm.log 2>&1; echo "exit $?" >> gpurun_out/probe_gemm.log; tail -4 gpurun_out/probe_gemm.log
timeout 120 $P attn ${PROBE_BATCH:-16} ${PROBE_ITERS:-10} > gpurun_out/probe_attn.log 2>&1; echo "exit $?" >> gpurun_out/probe_attn.log; tail -4 gpurun_out/probe_attn.log
# every k-loop variant of the plane-fed kernel (bitwise check against the fp32-fed kernel + timings)
for m in ${PROBE_APL_MODES:-}; do
  DDPO_APL_MODE=$m timeout 120 $P gemm2 ${PROBE_BATCH:-16} ${PROBE_ITERS:-10} > gpurun_out/probe_gemm2_mode$m.log 2>&1; echo "exit $?" >> gpurun_out/probe_gemm2_mode$m.log; tail -2 gpurun_out/probe_gemm2_mode$m.log
done
# the same A/B with cold weights (caches flushed before every timed launch, activations re-read): the in-model condition
for m in ${PROBE_COLD_MODES:-}; do
  PROBE_COLD=1 DDPO_APL_MODE=$m timeout 200 $P gemm2 ${PROBE_BATCH:-16} ${PROBE_ITERS:-10} > gpurun_out/probe_gemm2_cold_mode$m.log 2>&1; echo "exit $?" >> gpurun_out/probe_gemm2_cold_mode$m.log; tail -2 gpurun_out/probe_gemm2_cold_mode$m.log
done
for kv in ${PROBE_KNOBS:-}; do
  env $kv timeout 120 $P gemm ${PROBE_BATCH:-16} ${PROBE_ITERS:-10} > gpurun_out/probe_gemm_${kv//[^A-Za-z0-9_=]/_}.log 2>&1
done
if [ -n "${PROBE_TRAIN_BATCH:-}" ]; then      # the same GEMM shapes at the U-Net batch of one PPO micro-step (2 samples x CFG = 4)
  timeout 120 $P gemm $PROBE_TRAIN_BATCH ${PROBE_ITERS:-10} > gpurun_out/probe_gemm_b$PROBE_TRAIN_BATCH.log 2>&1
fi
