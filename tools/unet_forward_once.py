#!/usr/bin/env python
"""Runs N eager SD-1.5 U-Net forwards at batch 16 (the sampling step of the headline bench: 8 images x CFG, planes on) and prints
the algorithmic bytes / FLOPs of its GEMM / conv launches — the population tools/pmc_unet_traffic.sh collects FETCH_SIZE / WRITE_SIZE
over, so the counter traffic and the algorithmic bytes of bench.py's `roofline` object describe THE SAME launches."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpo_amd import lib as L
from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
L.DATAPATH = L.shipped_datapath()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 16
unet = UNet2DCondition(UNetConfig.named("sd15"), "cuda")
unet.params.init_synthetic(0)
unet.params.pack_bf16(bwd=False)
x = torch.randn(B, 4, 64, 64, device="cuda"); t = torch.full((B,), 481, dtype=torch.int32, device="cuda"); c = torch.randn(B, 77, 768, device="cuda")
L.PROFILE = []
for _ in range(N):
    unet(x, t, c, cfg_dup=False)
torch.cuda.synchronize()
recs = L.PROFILE; L.PROFILE = None
ms = sum(r[0].elapsed_time(r[1]) for r in recs)
print(json.dumps({"forwards": N, "gemm_launches": len(recs), "algorithmic_bytes_per_launch": sum(r[4] for r in recs) / len(recs),
                  "algorithmic_gflop_per_launch": sum(r[2] for r in recs) / len(recs) / 1e9, "event_ms_per_launch": ms / len(recs)}))
