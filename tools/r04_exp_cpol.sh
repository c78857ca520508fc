#!/bin/bash
# Cache-policy experiment of the short-reduction GEMMs (tools/native `make exp`): is M=65536,K=320,N=320 memory-system bound in steady state
# (timed over 3 / 10 / 50 back-to-back launches), and do a streaming hint on the activation LDS-DMA / nontemporal output stores help?
cd "$(dirname "$0")/native" || exit 1
export PROBE_WKBLK=1
for it in 3 10 50; do echo "== base, d0, iters $it"; PROBE_ONLY=d0 timeout 60 ./kernel_probe gemm2 16 $it | tail -1; done
for v in "" _nta _nto _ntao; do
  echo "== variant kernel_probe$v (iters 20)"
  timeout 200 ./kernel_probe$v gemm2 16 20 | grep -v "^#"
done
