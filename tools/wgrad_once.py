#!/usr/bin/env python
"""A few launches of the bf16x3 weight-gradient kernel on one U-Net layer (conv 3x3 320->320 @ 64x64, U-Net batch 64 = the fused train
step's level-0 shape; activation from planes, dY fp32 — the shipped configuration) for tools/pmc_wgrad.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpo_amd import lib as L
L.DATAPATH = "bf16x3"
B, H, C = int(os.environ.get("WG_B", "64")), 64, 320
x = torch.randn(B * H * H, C, device="cuda")
dy = torch.randn(B * H * H, C, device="cuda")
dw = torch.zeros(3, 3, C, C, device="cuda")
xp = L.split_planes(x)
for _ in range(3):
    L.conv2d_wgrad(xp, dy, dw, B, H, H, C, C, 3)
for _ in range(2):
    L.conv2d_wgrad(x, dy, dw, B, H, H, C, C, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    L.conv2d_wgrad(xp, dy, dw, B, H, H, C, C, 3)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"wgrad conv3x3 {C}->{C} @{H}^2 B{B}: {ms:.3f} ms  {2.0 * B * H * H * 9 * C * C / ms / 1e9:.1f} TF (A from planes)")

# a second layer that stays on the 128x128 kernel (conv 3x3 640->640 @ 32^2): the counters of both tiles come from one process
C2, H2 = 640, 32
x2 = torch.randn(B * H2 * H2, C2, device="cuda"); dy2 = torch.randn(B * H2 * H2, C2, device="cuda"); dw2 = torch.zeros(3, 3, C2, C2, device="cuda")
xp2 = L.split_planes(x2)
for _ in range(3):
    L.conv2d_wgrad(xp2, dy2, dw2, B, H2, H2, C2, C2, 3)
torch.cuda.synchronize()
