#!/usr/bin/env python
"""Host-side sweep: seconds per oracle step (SD-1.5 U-Net forward on a batch of 2 at 64x64 latents, fp32 — what every full-size parity test
waits for) and per forward + backward at 32x32, against torch's CPU thread count.  Picks the default of tests/conftest.py.
    python tools/oracle_threads.py [threads ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet as OU

counts = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, os.cpu_count()]
op = OU.init_params(OU.unet_param_shapes(OU.SD15), seed=0)
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 4, 64, 64, generator=g)
xs = torch.randn(2, 4, 32, 32, generator=g)
ctx = torch.randn(2, 77, 768, generator=g)
t = torch.full((2,), 481, dtype=torch.int32)
print(f"host: {os.cpu_count()} hardware threads, torch default {torch.get_num_threads()}")
for n in counts:
    torch.set_num_threads(n)
    with torch.no_grad():
        OU.unet_forward(op, OU.SD15, xs, t, ctx)
        t0 = time.perf_counter()
        OU.unet_forward(op, OU.SD15, x, t, ctx)
        fwd = time.perf_counter() - t0
    leaves = {k: v.clone().requires_grad_(True) for k, v in op.items()}
    t0 = time.perf_counter()
    OU.unet_forward(leaves, OU.SD15, xs, t, ctx).square().mean().backward()
    fb = time.perf_counter() - t0
    print(f"threads {n:4d}: forward 64x64 B=2 {fwd:6.2f} s | forward + backward 32x32 B=2 {fb:6.2f} s", flush=True)
