"""Plane-fed bf16x3 GEMM / conv (LDS-DMA operands) and the plane-emitting GroupNorm / LayerNorm output stages.

Contract under test: an activation written as bf16 hi / lo planes by a producer and consumed by
`ddpo_gemm_conv_fwd_bf16_planes` gives BIT-IDENTICAL results to the fp32 tensor consumed by `ddpo_gemm_conv_fwd_bf16`
(same tiles, same k order, same MFMA passes; the planes hold exactly the split the fp32-fed loader computes on the fly).
The U-Net / VAE sampling forward therefore does not change by a single bit when `lib.PLANES` is switched on, and the
training forward (always fp32-fed: its backward needs the fp32 activations) keeps matching the sampler bit for bit.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ddpo_amd import lib as L

DEV = "cuda"


@pytest.fixture(autouse=True)
def _planes_on(monkeypatch):
    """The plane-fed path is opt-in (DDPO_PLANES=1); these tests exercise it whatever the environment says."""
    monkeypatch.setattr(L, "PLANES", True)
    monkeypatch.setattr(L, "PLANES_ALL", True)       # plane-feed every eligible layer (the model's speed rule planes_pay() skips small ones)


def _bf16x3():
    L.DATAPATH = "bf16x3"


@pytest.mark.parametrize("kblocked", [False, True])
def test_split_planes_is_the_loader_split(kblocked, monkeypatch):
    monkeypatch.setattr(L, "A_KBLOCKED", kblocked)
    x = torch.randn(257, 96, device=DEV) * torch.logspace(-6, 4, 96, device=DEV)
    pl = L.split_planes(x)
    assert pl.kblocked == kblocked and pl.ld == (0 if kblocked else 96) and tuple(pl.hi.shape) == ((3, 257, 32) if kblocked else (257, 96))
    hi_ref = x.bfloat16()                                             # round-to-nearest-even, like v_cvt_pk_bf16_f32
    assert torch.equal(pl.plane("hi").view(torch.bfloat16), hi_ref)
    lo_ref = (x - hi_ref.float()).bfloat16()
    assert torch.equal(pl.plane("lo").view(torch.bfloat16), lo_ref)
    assert float(((pl.float() - x).abs() / x.abs().clamp_min(1e-30)).max()) < 2.0 ** -15


@pytest.mark.parametrize("B,HW,C,G,silu", [(2, 64, 64, 32, True), (3, 49, 96, 32, False), (1, 1024, 320, 32, True),
                                           (2, 25, 36, 4, True)])        # C % 8 != 0: the unpaired 8-byte plane stores
def test_groupnorm_planes_equal_split_of_fp32_result(B, HW, C, G, silu):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B * HW, C, device=DEV, generator=g) * 3 + 0.5
    gamma, beta = torch.randn(C, device=DEV, generator=g), torch.randn(C, device=DEV, generator=g)
    y = L.groupnorm(x, B, HW, gamma, beta, G, 1e-5, silu)
    pl = L.groupnorm(x, B, HW, gamma, beta, G, 1e-5, silu, planes=True)
    ref = L.split_planes(y)
    assert torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo)


@pytest.mark.parametrize("rows,C", [(77, 64), (1024, 320), (130, 1280), (10, 36)])
def test_layernorm_planes_equal_split_of_fp32_result(rows, C):
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(rows, C, device=DEV, generator=g) * 2 - 0.3
    gamma, beta = torch.randn(C, device=DEV, generator=g), torch.randn(C, device=DEV, generator=g)
    ref = L.split_planes(L.layernorm(x, gamma, beta))
    pl = L.layernorm(x, gamma, beta, planes=True)
    assert torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo)


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride,ups", [
    (2, 16, 16, 64, 96, 3, 1, False),      # 128x64 tiles, ragged N
    (4, 32, 32, 320, 320, 3, 1, False),    # 128x320 tiles
    (2, 16, 16, 128, 128, 3, 1, False),    # 128x128 / 128x64
    (1, 8, 8, 1280, 1280, 3, 1, False),    # split-K
    (2, 16, 16, 64, 64, 3, 2, False), (2, 8, 8, 64, 64, 3, 1, True), (3, 12, 20, 96, 160, 1, 1, False)])
def test_conv_planes_bit_identical(B, H, W, Cin, Cout, ks, stride, ups):
    _bf16x3()
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B * H * W, Cin, device=DEV, generator=g)
    w = torch.randn(ks, ks, Cin, Cout, device=DEV, generator=g) / (ks * ks * Cin) ** 0.5
    b = torch.randn(Cout, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    assert L.planes_ok(w, Cin, B * H * W)
    y0, OH, OW = L.conv2d(x, w, b, B, H, W, Cin, Cout, ks, stride=stride, upsample=ups)
    res = torch.randn_like(y0)
    rb = torch.randn(B, Cout, device=DEV, generator=g)
    y1, _, _ = L.conv2d(x, w, b, B, H, W, Cin, Cout, ks, stride=stride, upsample=ups, residual=res, rowbias=rb, rows_per_batch=OH * OW)
    pl = L.split_planes(x)
    z0, _, _ = L.conv2d(pl, w, b, B, H, W, Cin, Cout, ks, stride=stride, upsample=ups)
    z1, _, _ = L.conv2d(pl, w, b, B, H, W, Cin, Cout, ks, stride=stride, upsample=ups, residual=res, rowbias=rb, rows_per_batch=OH * OW)
    assert torch.equal(y0, z0) and torch.equal(y1, z1)


@pytest.mark.parametrize("M,K,N", [(77, 64, 96), (4096, 320, 320), (1024, 640, 5120), (300, 1280, 1280), (64, 5120, 1280), (5, 32, 8)])
def test_linear_planes_bit_identical(M, K, N):
    _bf16x3()
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(K, N, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    res = torch.randn(M, N, device=DEV, generator=g)
    y = L.linear(x, w, b, residual=res)
    z = L.linear(L.split_planes(x), w, b, residual=res)
    assert torch.equal(y, z)


def test_linear_geglu_planes_bit_identical():
    _bf16x3()
    g = torch.Generator(device=DEV).manual_seed(5)
    M, K, F = 1000, 320, 1280
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(K, 2 * F, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(2 * F, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    assert L.pack_weights_geglu(w, b)
    y = L.linear_geglu(x, w)
    z = L.linear_geglu(L.split_planes(x), w)
    assert y is not None and torch.equal(y, z)


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride,ups,fed", [
    (16, 64, 64, 320, 320, 1, 1, False, "planes"),   # 256x320 tall tiles (plane-fed only)
    (4, 32, 32, 320, 320, 3, 1, False, "fp32"),      # 128x320 tiles, fp32-fed
    (4, 32, 32, 320, 640, 3, 1, False, "planes"),    # 128x320 tiles, plane-fed
    (2, 16, 16, 128, 128, 3, 1, False, "fp32"),      # 128x128 / 128x64
    (1, 8, 8, 1280, 1280, 3, 1, False, "planes"),    # split-K: the reduce kernel writes the planes
    (2, 16, 16, 64, 64, 3, 2, False, "fp32"), (2, 8, 8, 64, 64, 3, 1, True, "planes")])
def test_gemm_output_stage_emits_the_loader_split_of_its_result(B, H, W, Cin, Cout, ks, stride, ups, fed):
    """ddpo_gemm_desc.out_hi / out_lo: the planes a GEMM's output stage writes are, bit for bit, split_planes() of the fp32 result
    it writes next to them (so a plane-fed consumer sees what an fp32-fed consumer would split on the fly) — and "only" mode
    (no fp32 tensor) writes the same planes."""
    _bf16x3()
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn(B * H * W, Cin, device=DEV, generator=g)
    w = torch.randn(ks, ks, Cin, Cout, device=DEV, generator=g) / (ks * ks * Cin) ** 0.5
    b = torch.randn(Cout, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    assert L.planes_out_ok(w, Cin, B * H * W, Cout)
    src = L.split_planes(x) if fed == "planes" else x
    y0, OH, OW = L.conv2d(src, w, b, B, H, W, Cin, Cout, ks, stride=stride, upsample=ups)
    res = torch.randn_like(y0)
    rb = torch.randn(B, Cout, device=DEV, generator=g)
    kw = dict(stride=stride, upsample=ups, residual=res, rowbias=rb, rows_per_batch=OH * OW)
    y1, _, _ = L.conv2d(src, w, b, B, H, W, Cin, Cout, ks, **kw)
    (z1, pl), _, _ = L.conv2d(src, w, b, B, H, W, Cin, Cout, ks, planes_out="both", **kw)
    only, _, _ = L.conv2d(src, w, b, B, H, W, Cin, Cout, ks, planes_out="only", **kw)
    ref = L.split_planes(y1)
    assert torch.equal(z1, y1)
    assert torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo)
    assert torch.equal(only.hi, ref.hi) and torch.equal(only.lo, ref.lo)


def test_linear_geglu_emits_planes_for_ff2():
    _bf16x3()
    g = torch.Generator(device=DEV).manual_seed(14)
    M, K, F = 1000, 320, 1280
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(K, 2 * F, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(2 * F, device=DEV, generator=g)
    w2 = torch.randn(F, K, device=DEV, generator=g) / F ** 0.5
    L.pack_weights(w, bwd=False)
    L.pack_weights(w2, bwd=False)
    assert L.pack_weights_geglu(w, b)
    y = L.linear_geglu(L.split_planes(x), w)
    pl = L.linear_geglu(L.split_planes(x), w, planes_out=True)
    ref = L.split_planes(y)
    assert isinstance(pl, L.Planes) and torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo)
    assert torch.equal(L.linear(pl, w2), L.linear(y, w2))                 # FF2 plane-fed == fp32-fed


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride,ups", [
    (4, 32, 32, 320, 320, 3, 1, False), (2, 16, 16, 128, 96, 3, 1, False), (2, 16, 16, 64, 64, 3, 2, False),
    (2, 8, 8, 64, 64, 3, 1, True), (3, 12, 20, 96, 160, 1, 1, False), (0, 777, 1, 320, 1280, 0, 1, False)])     # last: dense (M = 777)
@pytest.mark.parametrize("which", ["a", "dy", "both"])
def test_wgrad_from_planes_equals_wgrad_from_fp32(B, H, W, Cin, Cout, ks, stride, ups, which):
    """ddpo_gemm_conv_wgrad_bf16x3_planes: the weight gradient read from pre-split planes (forward input and / or dY) is the one
    the fp32-fed kernel computes — the planes ARE its on-the-fly split, only the fp32 atomics' order differs between launches."""
    _bf16x3()
    g = torch.Generator(device=DEV).manual_seed(21)
    if ks == 0:
        M, K, N = H, Cin, Cout
        x = torch.randn(M, K, device=DEV, generator=g)
        dy = torch.randn(M, N, device=DEV, generator=g)
        run = lambda a, b: L.linear_wgrad(a, b, torch.zeros(K, N, device=DEV))
    else:
        x = torch.randn(B * H * W, Cin, device=DEV, generator=g)
        pad = ks // 2
        VH, VW = (2 * H, 2 * W) if ups else (H, W)
        OH, OW = (VH + 2 * pad - ks) // stride + 1, (VW + 2 * pad - ks) // stride + 1
        dy = torch.randn(B * OH * OW, Cout, device=DEV, generator=g)
        run = lambda a, b: L.conv2d_wgrad(a, b, torch.zeros(ks, ks, Cin, Cout, device=DEV), B, H, W, Cin, Cout, ks, stride=stride, upsample=ups)
    ref = run(x, dy)
    a = L.split_planes(x) if which in ("a", "both") else x
    b = L.split_planes(dy) if which in ("dy", "both") else dy
    got = run(a, b)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-6 * scale + 1e-7


def test_plane_emitting_stage_rejected_off_the_buffer_addressed_kernels():
    _bf16x3()
    x = torch.randn(64, 40, device=DEV)                 # K % 32 != 0 -> generic loader: no plane-emitting output stage
    w = torch.randn(40, 64, device=DEV)
    L.pack_weights(w, bwd=False)
    assert not L.planes_out_ok(w, 40, 64, 64)
    with pytest.raises(L.DdpoHipError):
        L.linear(x, w, planes_out="both")
    L.DATAPATH = "fp32"
    w3 = torch.randn(64, 64, device=DEV)
    with pytest.raises(L.DdpoHipError):
        L.linear(torch.randn(64, 64, device=DEV), w3, planes_out="both")


def test_planes_rejected_where_the_fp32_entry_must_be_used():
    _bf16x3()
    x = torch.randn(64, 40, device=DEV)                 # K % 32 != 0
    w = torch.randn(40, 64, device=DEV)
    L.pack_weights(w, bwd=False)
    assert not L.planes_ok(w, 40, 64)
    with pytest.raises(L.DdpoHipError):
        L.linear(L.split_planes(x), w)
    w2 = torch.randn(64, 64, device=DEV)                # weight planes not registered
    assert not L.planes_ok(w2, 64, 64)
    w3 = torch.randn(64, 64, device=DEV)
    L.pack_weights(w3, bwd=False)
    assert L.planes_ok(w3, 64, 1000) and not L.planes_ok(w3, 64, 1 << 24)      # 31-bit byte offsets (buf_path_ok)
    L.DATAPATH = "fp32"
    assert not L.planes_ok(w3, 64, 1000)


@pytest.mark.parametrize("model", ["tiny", "tiny21"])
def test_unet_and_vae_forward_unchanged_by_planes(model, monkeypatch):
    """Sampling forward with plane-fed GEMMs == without, bit for bit; the training forward (tape) never uses planes."""
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    from ddpo_amd.models.vae import VAEDecoder, VAEConfig
    _bf16x3()
    cfg = UNetConfig.named(model)
    unet = UNet2DCondition(cfg, DEV)
    unet.params.init_synthetic(0)
    unet.params.pack_bf16()
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(4, 4, 16, 16, device=DEV, generator=g)
    t = torch.tensor([981, 21, 481, 1], dtype=torch.int32, device=DEV)
    ctx = torch.randn(4, 77, cfg.cross_attention_dim, device=DEV, generator=g)
    monkeypatch.setattr(L, "PLANES", True)
    counter = {"n": 0}
    real = L.gemm_conv

    def counting(src, *a, **k):
        counter["n"] += isinstance(src, L.Planes)
        return real(src, *a, **k)
    monkeypatch.setattr(L, "gemm_conv", counting)
    y_pl = unet(x, t, ctx)
    assert counter["n"] > 20                                  # the plane-fed path really ran
    tape = []
    counter["n"] = 0
    monkeypatch.setattr(L, "TRAIN_PLANES", False)
    y_train = unet.forward(x, t, ctx, tape=tape)
    assert counter["n"] == 0                                  # DDPO_TRAIN_PLANES=0: the training forward keeps fp32 activations
    monkeypatch.setattr(L, "TRAIN_PLANES", True)
    y_train_pl = unet.forward(x, t, ctx, tape=[])
    assert counter["n"] > 10 and torch.equal(y_train_pl, y_train)      # default: norm outputs as planes, same bits
    counter["n"] = 0
    monkeypatch.setattr(L, "PLANES", False)
    y_fp = unet(x, t, ctx)
    assert counter["n"] == 0
    assert torch.equal(y_pl, y_fp) and torch.equal(y_pl, y_train)
    vae = VAEDecoder(VAEConfig.named("tiny"), DEV)
    vae.params.init_synthetic(1)
    vae.params.pack_bf16(bwd=False)
    lat = torch.randn(2, 4, 8, 8, device=DEV, generator=g)
    img_fp = vae.decode(lat)
    monkeypatch.setattr(L, "PLANES", True)
    img_pl = vae.decode(lat)
    assert torch.equal(img_fp, img_pl)


@pytest.mark.parametrize("batch,wkblk", [("4", "0"), ("4", "1"), ("16", "0"), ("16", "1")])
def test_kernel_probe_plane_fed_kernels_bit_identical(batch, wkblk):
    """tools/native/kernel_probe gemm2 compares the plane-fed kernels with the fp32-fed ones output by output on 26 layer shapes
    (conv borders, strides, upsampling, ragged M / N, split-K) through the C ABI alone (no torch) and exits non-zero on any mismatch.
    Batch 4 keeps every layer on the 128-row tiles (three weight stages, staggered requests); batch 16 moves the 64x64-level layers to
    the 256x320 tile (rotated schedule).  Both forward weight-plane layouts (row-major, k-blocked).  The k-loop variants that lost
    their measurements (two weight stages, no stagger, plain tall loop, spread requests, s_setprio, four-wave tiles) are no longer in
    the library (profiles/r02_ab_apl_mode.log, profiles/r03_probe_kloop.log)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "native", "kernel_probe")
    if not os.path.exists(exe):          # build it here rather than skip: on the GPU box this test is the tall tile's bit-identity evidence
        mk = subprocess.run(["make", "-C", os.path.join(root, "tools", "native")], capture_output=True, text=True, timeout=900)
        assert mk.returncode == 0 and os.path.exists(exe), "tools/native/kernel_probe is missing and could not be built:\n" + mk.stdout[-2000:] + mk.stderr[-2000:]
    out = subprocess.run([exe, "gemm2", batch, "2"], env=dict(os.environ, PROBE_WKBLK=wkblk), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "FAIL" not in out.stdout and out.stdout.count("bit-identical") >= 20


@pytest.mark.parametrize("case", [("dense", 4096, 320, 320), ("dense", 1000, 1280, 640), ("dense", 77, 64, 96),
                                  ("conv", 2, 32, 320, 320, 3, 1, False), ("conv", 2, 16, 64, 96, 3, 2, False), ("conv", 1, 16, 64, 64, 3, 1, True),
                                  ("conv", 1, 64, 320, 320, 1, 1, False)])
def test_kblocked_activation_planes_are_bit_identical_to_row_major(monkeypatch, case):
    """Activation planes stored k-blocked (C / 32, rows, 32) (plane row stride 0 in the C ABI, ABI v6) against row-major (rows, C): every
    producer (split_planes, GroupNorm, LayerNorm, the plane-emitting GEMM output stage) and every consumer (plane-fed forward GEMM /
    conv with stride, padding and nearest-2x upsampling in the gather; the weight gradient from activation planes, dY planes, both)
    must see the same values: forward outputs, emitted planes and weight gradients bit-identical between the two storages."""
    _bf16x3()
    res = []
    for kb in (False, True):
        monkeypatch.setattr(L, "A_KBLOCKED", kb)
        L.PACKED.clear()
        g = torch.Generator().manual_seed(3)
        if case[0] == "dense":
            _, M, K, N = case
            x = torch.randn(M, K, generator=g).to(DEV)
            w = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
            b = torch.randn(N, generator=g).to(DEV)
            dy = torch.randn(M, N, generator=g).to(DEV)
            L.pack_weights(w)
            gam, bet = torch.randn(K, generator=g).to(DEV), torch.randn(K, generator=g).to(DEV)
            pl = L.layernorm(x, gam, bet, planes=True)
            assert pl.kblocked == kb
            out, opl = L.linear(pl, w, b, planes_out="both")
            dw1, dw2, dw3 = (torch.zeros_like(w) for _ in range(3))
            L.gemm_wgrad(pl, dy, dw1, M=M, N=N, K=K)
            L.gemm_wgrad(pl, L.split_planes(dy), dw2, M=M, N=N, K=K)
            L.gemm_wgrad(x, L.split_planes(dy), dw3, M=M, N=N, K=K)
            res.append([out, opl.float(), pl.float(), dw1, dw2, dw3])
        else:
            _, B, H, Cin, Cout, ks, stride, ups = case
            x = torch.randn(B * H * H, Cin, generator=g).to(DEV)
            w = (torch.randn(ks, ks, Cin, Cout, generator=g) / (ks * ks * Cin) ** 0.5).to(DEV)
            b = torch.randn(Cout, generator=g).to(DEV)
            L.pack_weights(w)
            gam, bet = torch.randn(Cin, generator=g).to(DEV), torch.randn(Cin, generator=g).to(DEV)
            pl = L.groupnorm(x, B, H * H, gam, bet, 32, 1e-5, True, planes=True)
            assert pl.kblocked == kb
            (out, opl), OH, OW = L.conv2d(pl, w, b, B, H, H, Cin, Cout, ks, stride=stride, upsample=ups, planes_out="both")
            dy = torch.randn(B * OH * OW, Cout, generator=g).to(DEV)
            dw1, dw2 = torch.zeros_like(w), torch.zeros_like(w)
            L.conv2d_wgrad(pl, dy, dw1, B, H, H, Cin, Cout, ks, stride=stride, upsample=ups)
            L.conv2d_wgrad(pl, L.split_planes(dy), dw2, B, H, H, Cin, Cout, ks, stride=stride, upsample=ups)
            res.append([out, opl.float(), pl.float(), dw1, dw2])
    for i, (a, b_) in enumerate(zip(*res)):
        if i >= 3:          # weight gradients: fp32 atomics over the pixel splits -> equal up to the summation order
            assert float((a - b_).abs().max()) <= 2e-6 * float(b_.abs().max()) + 1e-7
        else:
            assert torch.equal(a, b_), i


@pytest.mark.parametrize("shape", [(3, 3, 64, 96), (3, 3, 320, 320), (1, 1, 96, 40), (640, 320), (100, 72)])
def test_dgrad_planes_packed_straight_from_w_equal_the_planes_of_the_flipped_transposed_copy(shape):
    """ADVICE r04 / ABI v13: ddpo_pack_weights_bf16_kblocked_dgrad writes the data-gradient operand W'[tap' * Cout + co][ci] =
    w[taps - 1 - tap'][ci][co] from the forward kernel directly; rounds 3-4 materialised the flipped / transposed fp32 copy with torch after
    every optimizer update and packed that.  Same planes bit for bit, ragged K' (zero padded to whole 32-blocks) included."""
    import ctypes
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(*shape, generator=g).to("cuda")
    wt = w.flip(0, 1).permute(0, 1, 3, 2).contiguous() if w.dim() == 4 else w.t().contiguous()
    Nd = wt.shape[-1]
    Kd = wt.numel() // Nd
    Kp = (Kd + 31) // 32 * 32
    mk = lambda: torch.full((Nd, Kp), 0x5555, dtype=torch.int16, device="cuda")
    rh, rl, nh, nl = mk(), mk(), mk(), mk()
    lib = L.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.ddpo_pack_weights_bf16_kblocked(p(wt), Kd, Nd, p(rh), p(rl), None) == 0
    taps = shape[0] * shape[1] if len(shape) == 4 else 1
    cin, cout = (shape[2], shape[3]) if len(shape) == 4 else shape
    assert lib.ddpo_pack_weights_bf16_kblocked_dgrad(p(w), taps, cin, cout, p(nh), p(nl), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(nh, rh) and torch.equal(nl, rl)
