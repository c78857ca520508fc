#!/usr/bin/env python
"""Timing ablations of the 128x128 bf16x3 GEMM k-loop (debug entry ddpo_debug_gemm_ablate; results are wrong by design
for mode != 0).  mode bits: 1 = no global loads in the loop, 2 = no split + LDS stores, 4 = no barriers, 8 = no fragment reads,
16 = no epilogue stores, 32 = no prologue loads."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpo_amd import lib as L
L.DATAPATH = "bf16x3"
dev = "cuda"
lib = L.load()


MODES = (0, 16, 15, 31, 47, 63) if len(sys.argv) > 1 and sys.argv[1] == "short" else (0, 1, 2, 6, 7, 8, 15)


def run(M, K, N, conv=None):
    if conv:
        B, H, Cin = conv
        x = torch.randn(B * H * H, Cin, device=dev); w = torch.randn(3, 3, Cin, N, device=dev) * 0.02
    else:
        x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.02
    L.pack_weights(w)
    ent = L.PACKED[w.data_ptr()]
    hi, lo, ldw = ent["fwd"]      # row-major planes: run with DDPO_W_KBLOCKED=0
    out = torch.empty(M, N, device=dev)
    d = L.GemmDesc()
    d.src = x.data_ptr(); d.ld_src = x.shape[1]; d.out = out.data_ptr(); d.ld_out = N; d.alpha = 1.0
    d.M, d.N, d.K = M, N, K
    if conv:
        d.ksize, d.stride, d.pad, d.upsample, d.B, d.H, d.W, d.Cin, d.OH, d.OW = 3, 1, 1, 0, B, H, H, Cin, H, H
    res = []
    for mode in MODES:
        f = lambda: lib.ddpo_debug_gemm_ablate(ctypes.byref(d), hi.data_ptr(), lo.data_ptr(), ldw, mode, None)
        for _ in range(2): assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(f"m{mode}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:6.1f} TF")
    print(f"M={M} K={K} N={N} conv={conv}: " + " | ".join(res))


import sys
MODES = (0, 16, 15, 31, 47, 63) if len(sys.argv) > 1 and sys.argv[1] == "short" else (0, 1, 2, 6, 7, 8, 15)
if len(sys.argv) > 1 and sys.argv[1] == "short":
    run(65536, 320, 2560)
    run(65536, 320, 384)
    run(16384, 640, 640)
    run(4096, 1280, 1280)
    run(65536, 1280, 384)
else:
    run(4096, 1280, 10240)
    run(16384, 2560, 640)
    run(16 * 32 * 32, 9 * 640, 640, conv=(16, 32, 640))
    run(16 * 64 * 64, 9 * 640, 640, conv=(16, 64, 640))
