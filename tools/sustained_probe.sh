#!/bin/bash
# Is the sampling bench clock / power limited?  Runs the fp32-fed vs plane-fed comparison with LONG timing loops (seconds of
# back-to-back launches per layer instead of the probe's millisecond bursts) while logging sclk / power every 200 ms.
#   gpurun --timeout 240 -- 'bash tools/sustained_probe.sh'
# Read: if the x-ratio of `planes` over `fp32-fed` shrinks towards 1.0 as iters grows and sclk sits well below its burst
# value, the chip is throttling on matrix activity and kernel-level gains will not show up end to end.
mkdir -p gpurun_out
P=tools/native/kernel_probe
( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/smi_trace.csv 2>&1 &
SMI=$!
for it in 10 300 3000; do
  DDPO_APL_MODE=${DDPO_APL_MODE:-6} timeout 100 $P gemm2 16 $it 2>&1 | grep -E "conv 3x3 s1 up0 +(320|960|1920)->" > gpurun_out/sustained_iters$it.log
  echo "== iters $it"; cat gpurun_out/sustained_iters$it.log | cut -c1-170
done
kill $SMI
tail -5 gpurun_out/smi_trace.csv
