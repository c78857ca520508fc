#!/usr/bin/env python
"""Per-launch breakdown of the GEMM/conv kernels of one SD-1.5 U-Net forward (batch 16 = 8 images x CFG)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpo_amd import lib as L
from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
L.DATAPATH = os.environ.get("DDPO_DATAPATH", "bf16x3")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
unet = UNet2DCondition(UNetConfig.named("sd15"), "cuda")
unet.params.init_synthetic(0)
if L.DATAPATH != "fp32":
    unet.params.pack_bf16(bwd=False)
x = torch.randn(B, 4, 64, 64, device="cuda"); t = torch.full((B,), 481, dtype=torch.int32, device="cuda"); c = torch.randn(B, 77, 768, device="cuda")
# monkeypatch gemm_conv to record shapes
orig = L.gemm_conv
shapes = []
def rec(src, w, *, M, N, K, conv=None, **kw):
    shapes.append((M, N, K, (conv or {}).get("ksize", 0), (conv or {}).get("stride", 1), (conv or {}).get("upsample", 0)))
    return orig(src, w, M=M, N=N, K=K, conv=conv, **kw)
L.gemm_conv = rec
orig_geglu = L.linear_geglu
def rec_geglu(x, w, out=None):
    r = orig_geglu(x, w, out=out)
    if r is not None:
        shapes.append((x.shape[0], w.shape[1], x.shape[1], -1, 1, 0))      # ks = -1 marks the fused FF1 + GEGLU launch
    return r
L.linear_geglu = rec_geglu
unet(x, t, c); torch.cuda.synchronize()
shapes.clear(); L.PROFILE = []
unet(x, t, c); torch.cuda.synchronize()
recs = L.PROFILE; L.PROFILE = None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for sh, r in zip(shapes, recs):
    a = agg[sh]; a[0] += 1; a[1] += r[0].elapsed_time(r[1]); a[2] += r[2]
tot = sum(a[1] for a in agg.values())
print(f"total gemm/conv ms {tot:.2f} for {len(recs)} launches; {sum(a[2] for a in agg.values())/tot/1e9:.1f} TF avg")
for sh, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"M={sh[0]:6d} N={sh[1]:5d} K={sh[2]:6d} ks={sh[3]} s={sh[4]} up={sh[5]} x{a[0]:3d}: {a[1]:7.2f} ms ({100*a[1]/tot:4.1f}%) {a[2]/a[1]/1e9:6.1f} TF")
