#!/usr/bin/env python
"""Per-launch breakdown of the GEMM/conv kernels of one SD-1.5 U-Net forward (batch 16 = 8 images x CFG), in the model
(cold weights, producers' outputs as inputs) — the counterpart of tools/native/kernel_probe's isolated layers.

    python tools/unet_gemm_breakdown.py [batch]            fp32-fed kernels
    python tools/unet_gemm_breakdown.py [batch] --ab       fp32-fed vs plane-fed (lib.PLANES) side by side, per layer shape
"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpo_amd import lib as L
from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
L.DATAPATH = L.shipped_datapath()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
AB = "--ab" in sys.argv
B = int(args[0]) if args else 16
MODEL = os.environ.get("MODEL", "sd15")              # MODEL=sd21: the SD-2.1 U-Net at 96x96 latents (BASELINE configs[4])
HW, CTX = (96, 1024) if MODEL == "sd21" else (64, 768)
unet = UNet2DCondition(UNetConfig.named(MODEL), "cuda")
unet.params.init_synthetic(0)
if L.DATAPATH != "fp32":
    unet.params.pack_bf16(bwd=False)
x = torch.randn(B, 4, HW, HW, device="cuda"); t = torch.full((B,), 481, dtype=torch.int32, device="cuda"); c = torch.randn(B, 77, CTX, device="cuda")
# monkeypatch gemm_conv / linear_geglu to record shapes (+ whether the activation came as planes)
orig = L.gemm_conv
shapes = []
def rec(src, w, *, M, N, K, conv=None, **kw):
    shapes.append((M, N, K, (conv or {}).get("ksize", 0), (conv or {}).get("stride", 1), (conv or {}).get("upsample", 0), isinstance(src, L.Planes)))
    return orig(src, w, M=M, N=N, K=K, conv=conv, **kw)
L.gemm_conv = rec
orig_geglu = L.linear_geglu
def rec_geglu(x, w, out=None, **kw):
    r = orig_geglu(x, w, out=out, **kw)
    if r is not None:
        shapes.append((x.shape[0], w.shape[1], x.shape[1], -1, 1, 0, isinstance(x, L.Planes)))      # ks = -1 marks the fused FF1 + GEGLU launch
    return r
L.linear_geglu = rec_geglu


def measure(planes, reps=3, planes_out=True):
    L.PLANES = planes
    L.PLANES_OUT = planes_out
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, False])
    unet(x, t, c); torch.cuda.synchronize()
    for _ in range(reps):
        shapes.clear(); L.PROFILE = []
        unet(x, t, c); torch.cuda.synchronize()
        recs = L.PROFILE; L.PROFILE = None
        assert len(recs) == len(shapes)
        for sh, r in zip(shapes, recs):
            a = agg[sh[:6]]; a[0] += 1; a[1] += r[0].elapsed_time(r[1]); a[2] += r[2]; a[3] = a[3] or sh[6]
    for a in agg.values():
        a[0] //= reps; a[1] /= reps; a[2] /= reps
    return agg


base = measure(False)
tot = sum(a[1] for a in base.values())
print(f"fp32-fed: total gemm/conv ms {tot:.2f} for {sum(a[0] for a in base.values())} launches; {sum(a[2] for a in base.values())/tot/1e9:.1f} TF avg")
if not AB:
    for sh, a in sorted(base.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"M={sh[0]:6d} N={sh[1]:5d} K={sh[2]:6d} ks={sh[3]} s={sh[4]} up={sh[5]} x{a[0]:3d}: {a[1]:7.2f} ms ({100*a[1]/tot:4.1f}%) {a[2]/a[1]/1e9:6.1f} TF")
else:
    pl0 = measure(True, planes_out=False)
    t0 = sum(a[1] for a in pl0.values())
    print(f"planes, no plane-emitting output stages: total gemm/conv ms {t0:.2f}; {sum(a[0] for a in pl0.values() if a[3])} launches plane-fed")
    pl = measure(True)
    tot2 = sum(a[1] for a in pl.values())
    print(f"planes  : total gemm/conv ms {tot2:.2f}; {sum(a[2] for a in pl.values())/tot2/1e9:.1f} TF avg; "
          f"{sum(a[0] for a in pl.values() if a[3])} launches plane-fed")
    best = sum(min(a[1], pl[sh][1]) if sh in pl else a[1] for sh, a in base.items())
    print(f"best of both per layer shape: {best:.2f} ms")
    for sh, a in sorted(base.items(), key=lambda kv: -kv[1][1])[:48]:
        b = pl.get(sh)
        if b is None:
            continue
        print(f"M={sh[0]:6d} N={sh[1]:5d} K={sh[2]:6d} ks={sh[3]:2d} s={sh[4]} up={sh[5]} x{a[0]:3d}: fp32-fed {a[1]:7.2f} ms {a[2]/a[1]/1e9:6.1f} TF | "
              f"{'planes ' if b[3] else 'fp32(*)'} {b[1]:7.2f} ms {b[2]/b[1]/1e9:6.1f} TF  x{a[1]/b[1]:.2f}")
