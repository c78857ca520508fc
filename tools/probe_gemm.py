#!/usr/bin/env python
"""Single-shape probe for rocprofv3 PMC runs: python tools/probe_gemm.py <conv|gemm> ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpo_amd import lib as L
dev = "cuda"
kind = sys.argv[1]
torch.manual_seed(0)
if kind == "conv":
    B, H, Cin, Cout = 16, 64, 320, 320
    x = torch.randn(B * H * H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.02; b = torch.randn(Cout, device=dev)
    if L.DATAPATH != "fp32": L.pack_weights(w)
    out, _, _ = L.conv2d(x, w, b, B, H, H, Cin, Cout, 3)
    for _ in range(3): L.conv2d(x, w, b, B, H, H, Cin, Cout, 3, out=out)
else:
    M, K, N = 4096, 1280, 10240
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.02
    if L.DATAPATH != "fp32": L.pack_weights(w)
    out = L.linear(x, w)
    for _ in range(3): L.linear(x, w, out=out)
torch.cuda.synchronize()
