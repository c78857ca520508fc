#!/bin/bash
# Is the short-reduction GEMM of the 64x64 level memory-system bound in steady state?  (a) the plain-copy yardstick, (b) the layer timed over
# 3 / 10 / 50 back-to-back launches (3 fit the Infinity Cache, 50 are steady state; the model is steady state).
cd "$(dirname "$0")/native" || exit 1
export PROBE_WKBLK=1
timeout 120 ./kernel_probe stream | grep -v "^#"
for c in d0 d2 d13 c10 c0; do
  for it in 3 10 50; do printf "iters %2d: " $it; PROBE_ONLY=$c timeout 60 ./kernel_probe gemm2 16 $it | grep -v "^#" | cut -c1-130; done
done
