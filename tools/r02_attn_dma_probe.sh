#!/bin/bash
# LDS-DMA attention forward (DDPO_ATTN_DMA=1; 2 = two query blocks per wave) against the packed-image kernel: torch-free probe + the attention tests
P=tools/native/kernel_probe
for rep in 1 2; do for dm in 0 1 2; do echo "== DMA=$dm"; DDPO_ATTN_DMA=$dm timeout 120 $P attn 16 20 2>&1 | grep -E "^attn" | head -3; done; done | tee gpurun_out/r02_probe_attn_dma.log
DDPO_ATTN_DMA=2 timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "attn or attention" 2>&1 | tail -3
