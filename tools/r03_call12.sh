#!/bin/bash
# round 3, GPU call: in-model per-layer A/B fp32-fed vs plane-fed with k-blocked weight planes (does the planes_pay rule still hold?)
mkdir -p gpurun_out; export TMPDIR=/tmp
DDPO_PLANES_ALL=1 timeout 600 python tools/unet_gemm_breakdown.py 16 --ab > gpurun_out/r03_gemm_breakdown_ab.md 2>&1; head -45 gpurun_out/r03_gemm_breakdown_ab.md | cut -c1-200
