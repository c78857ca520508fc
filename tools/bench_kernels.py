#!/usr/bin/env python
"""Micro-benchmarks of the dominant kernels on the shapes of one SD-1.5 U-Net forward at batch 16 (8 images x CFG)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpo_amd import lib as L

dev = "cuda"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv_case(B, H, Cin, Cout, ks, stride=1, ups=False):
    x = torch.randn(B * H * H, Cin, device=dev)
    w = torch.randn(ks, ks, Cin, Cout, device=dev) * 0.02
    if L.DATAPATH != "fp32":
        L.pack_weights(w)
    b = torch.randn(Cout, device=dev)
    out, OH, OW = L.conv2d(x, w, b, B, H, H, Cin, Cout, ks, stride=stride, upsample=ups)
    ms = timeit(lambda: L.conv2d(x, w, b, B, H, H, Cin, Cout, ks, stride=stride, upsample=ups, out=out))
    fl = 2.0 * B * OH * OW * Cout * ks * ks * Cin
    return ms, fl / ms / 1e6


def gemm_case(M, K, N):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(K, N, device=dev) * 0.02
    if L.DATAPATH != "fp32":
        L.pack_weights(w)
    out = L.linear(x, w)
    ms = timeit(lambda: L.linear(x, w, out=out))
    return ms, 2.0 * M * K * N / ms / 1e6


def attn_case(B, heads, Nq, Nk, d):
    C = heads * d
    q, k, v = (torch.randn(B * n, C, device=dev) for n in (Nq, Nk, Nk))
    o = L.attention(q, k, v, B, heads, Nq, Nk, d)
    ms = timeit(lambda: L.attention(q, k, v, B, heads, Nq, Nk, d, out=o))
    return ms, 4.0 * B * heads * Nq * Nk * d / ms / 1e6


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    print(f"batch {B} datapath {L.DATAPATH}")
    for (H, Cin, Cout, ks, st, up) in [(64, 320, 320, 3, 1, False), (32, 640, 640, 3, 1, False), (16, 1280, 1280, 3, 1, False),
                                       (8, 1280, 1280, 3, 1, False), (64, 960, 320, 3, 1, False), (32, 1920, 640, 3, 1, False),
                                       (16, 2560, 1280, 3, 1, False), (8, 2560, 1280, 3, 1, False), (32, 640, 640, 3, 1, True),
                                       (64, 320, 320, 3, 2, False), (64, 320, 320, 1, 1, False), (64, 4, 320, 3, 1, False),
                                       (64, 320, 4, 3, 1, False)]:
        ms, tf = conv_case(B, H, Cin, Cout, ks, st, up)
        print(f"conv {ks}x{ks} s{st} up{int(up)} {Cin:5d}->{Cout:5d} @{H:3d}^2 : {ms:8.3f} ms {tf/1e3:7.1f} TF")
    for (M, K, N) in [(B * 4096, 320, 2560), (B * 4096, 1280, 320), (B * 4096, 320, 320), (B * 1024, 640, 5120), (B * 1024, 2560, 640),
                      (B * 256, 1280, 10240), (B * 256, 5120, 1280), (B * 77, 768, 320), (B, 1280, 1280), (B * 64, 1280, 1280)]:
        ms, tf = gemm_case(M, K, N)
        print(f"gemm M={M:6d} K={K:5d} N={N:5d} : {ms:8.3f} ms {tf/1e3:7.1f} TF")
    for (Nq, Nk, d) in [(4096, 4096, 40), (4096, 77, 40), (1024, 1024, 80), (1024, 77, 80), (256, 256, 160), (64, 64, 160)]:
        ms, tf = attn_case(B, 8, Nq, Nk, d)
        print(f"attn Nq={Nq:5d} Nk={Nk:5d} d={d:3d} : {ms:8.3f} ms {tf/1e3:7.1f} TF")
