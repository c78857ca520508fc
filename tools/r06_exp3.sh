#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_exp3.log
{
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_bf16_planes.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15
echo "== sd21 bf16 breakdown, PLANES_ALL"; MODEL=sd21 DDPO_DATAPATH=bf16 DDPO_PLANES_ALL=1 timeout 600 python tools/unet_gemm_breakdown.py 16 --ab 2>&1 | grep -v amdgpu.ids
echo "== sd15 bf16 breakdown, PLANES_ALL"; MODEL=sd15 DDPO_DATAPATH=bf16 DDPO_PLANES_ALL=1 timeout 600 python tools/unet_gemm_breakdown.py 16 --ab 2>&1 | grep -v amdgpu.ids
} > $L 2>&1
tail -5 $L
