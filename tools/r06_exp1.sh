#!/bin/bash
# round 6, call 1: what bounds the f16mx k-loops — stream depth sweep (dma_bench2) and the timing build's ablations on the f16mx operator
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_exp1.log
{
echo "== dma_bench2"; timeout 300 tools/native/dma_bench2
for abl in 0 4 1 2 0 4; do
  echo "== ktmx DDPO_DBG_ABL=$abl"; DDPO_DBG_ABL=$abl timeout 300 tools/native/kernel_probe_timing ktmx 16 | grep -v '^#'
done
echo "== ktmx DDPO_MX_TALL=0 ABL=0"; DDPO_MX_TALL=0 DDPO_DBG_ABL=0 timeout 300 tools/native/kernel_probe_timing ktmx 16 | grep -v '^#'
echo "== ktmx DDPO_MX_TALL=0 ABL=4"; DDPO_MX_TALL=0 DDPO_DBG_ABL=4 timeout 300 tools/native/kernel_probe_timing ktmx 16 | grep -v '^#'
} > $L 2>&1
tail -5 $L
