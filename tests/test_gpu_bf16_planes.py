"""Single-pass bf16 on the plane-fed kernels (round 6, ABI v14: `ddpo_gemm_conv_fwd_bf16_planes` with both lo planes NULL).

BASELINE configs[4] (SD-2.1 768^2) names bf16 — one `v_mfma_f32_32x32x16_bf16` per product with fp32 accumulation, XLA's TPU default
precision, what `load_unet(dtype=bfloat16)` selects (datapath "bf16").  Until round 6 that datapath ran on the fp32-fed kernels only (fp32
activations converted in the loader).  Contract under test: the plane-fed form — hi planes of both operands by LDS-DMA, the 256 x 320 tile on a
four-stage LDS ring with a counted vmcnt (APL = 8) where its grid fills the chip, the 128-row tiles on the three-weight-stage loop — is
BIT-IDENTICAL to the fp32-fed single-pass kernel (hi = bf16(x) is the operand that kernel's loader forms; same k order per accumulator), never
touches the lo planes, and sits at bf16's error against float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ddpo_amd import lib as L

DEV = "cuda"


@pytest.fixture(autouse=True)
def _bf16_planes(monkeypatch):
    monkeypatch.setattr(L, "PLANES", True)
    monkeypatch.setattr(L, "PLANES_ALL", True)
    monkeypatch.setattr(L, "BF16_PLANES", True)
    monkeypatch.setattr(L, "DATAPATH", "bf16")
    yield
    L.PACKED.clear()


def _poison_lo(pl):
    """The single pass must not read the lo plane: fill it with NaN patterns."""
    pl.lo.fill_(0x7FC1 - 0x10000 if False else -63)          # 0xFFC1: a bf16 NaN pattern
    return pl


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride,ups", [
    (2, 16, 16, 64, 96, 3, 1, False),       # 128x64 tiles, ragged N
    (4, 32, 32, 320, 320, 3, 1, False),     # 128x320 tiles (24 weight pieces for 20: the padded stage)
    (2, 16, 16, 128, 128, 3, 1, False),     # 128x128 / 128x64
    (1, 8, 8, 1280, 1280, 3, 1, False),     # split-K
    (16, 64, 64, 320, 320, 3, 1, False),    # 256 tall tiles: the four-stage ring, 90 k-tiles
    (13, 64, 64, 320, 320, 1, 1, False),    # 208 tall tiles, 10 k-tiles
    (16, 32, 32, 640, 640, 3, 1, True),     # tall tiles over the up-sampling gather
    (1, 64, 64, 32, 320, 1, 1, False),      # ONE k-tile (shorter than the ring)
    (2, 16, 16, 64, 64, 3, 2, False), (3, 12, 20, 96, 160, 1, 1, False)])
def test_conv_single_pass_planes_bit_identical_to_fp32_fed(B, H, W, Cin, Cout, ks, stride, ups):
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
    pl = _poison_lo(L.split_planes(x))
    before = L.gemm_tile_launch_counts()["tall_256x320"]
    z0, _, _ = L.conv2d(pl, w, b, B, H, W, Cin, Cout, ks, stride=stride, upsample=ups)
    z1, _, _ = L.conv2d(pl, w, b, B, H, W, Cin, Cout, ks, stride=stride, upsample=ups, residual=res, rowbias=rb, rows_per_batch=OH * OW)
    torch.cuda.synchronize()
    tall = L.gemm_tile_launch_counts()["tall_256x320"] - before
    assert torch.equal(y0, z0) and torch.equal(y1, z1)
    M = B * OH * OW
    assert (tall == 2) == (Cout % 320 == 0 and M >= 200 * 256), (tall, M)
    # the datapath: one bf16 pass (8 significant bits per operand) against float64
    xi = x.view(B, H, W, Cin).permute(0, 3, 1, 2).double()
    if ups:
        xi = xi.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = torch.nn.functional.conv2d(xi, w.double().permute(3, 2, 0, 1), b.double(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1).reshape(M, Cout)
    err = float((z0.double() - ref).pow(2).mean().sqrt() / (ref - b.double()).pow(2).mean().sqrt())
    assert 1e-4 < err < 6e-3, err               # ~2e-3: single-pass bf16; three passes would sit at 1e-5


@pytest.mark.parametrize("M,K,N", [(77, 64, 96), (4096, 320, 320), (65536, 320, 320), (65500, 1280, 320), (1024, 640, 5120), (300, 1280, 1280), (64, 5120, 1280), (5, 32, 8)])
def test_linear_single_pass_planes_bit_identical_to_fp32_fed(M, K, N):
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(K, N, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    res = torch.randn(M, N, device=DEV, generator=g)
    y = L.linear(x, w, b, residual=res)
    z = L.linear(_poison_lo(L.split_planes(x)), w, b, residual=res)
    assert torch.equal(y, z)


def test_linear_geglu_single_pass_planes_bit_identical_and_emits_planes():
    g = torch.Generator(device=DEV).manual_seed(5)
    M, K, F = 1000, 320, 1280
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(K, 2 * F, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(2 * F, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    assert L.pack_weights_geglu(w, b)
    y = L.linear_geglu(x, w)
    pl = _poison_lo(L.split_planes(x))
    z = L.linear_geglu(pl, w)
    assert y is not None and torch.equal(y, z)
    zp = L.linear_geglu(pl, w, planes_out=1)           # the output stage writes the planes the (plane-fed) second feed-forward GEMM reads
    ref = L.split_planes(y)
    assert torch.equal(zp.hi, ref.hi) and torch.equal(zp.lo, ref.lo)


def test_exactly_one_lo_plane_is_refused():
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(256, 64, device=DEV, generator=g)
    w = torch.randn(64, 64, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    pl = L.split_planes(x)
    hi, lo, ldw, _ = L._bf16_route(w, 64, 64, None, False)
    d = L.GemmDesc()
    out = torch.empty(256, 64, device=DEV)
    d.out = out.data_ptr(); d.ld_out = 64; d.alpha = 1.0; d.M, d.N, d.K = 256, 64, 64
    d.w_layout = L.PACKED[w.data_ptr()].get("w_layout", 0)
    from ctypes import byref
    f = L.load().ddpo_gemm_conv_fwd_bf16_planes
    assert f(byref(d), L._p(pl.hi), None, pl.ld, L._p(hi), L._p(lo), ldw, None, 0, None) == -1
    assert f(byref(d), L._p(pl.hi), L._p(pl.lo), pl.ld, L._p(hi), None, ldw, None, 0, None) == -1
    assert f(byref(d), L._p(pl.hi), None, pl.ld, L._p(hi), None, ldw, None, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, L.linear(x, w))


@pytest.mark.parametrize("family,ctx_dim", [("tiny", 64), ("tiny21", 96)])
def test_unet_forward_unchanged_by_single_pass_planes(family, ctx_dim, monkeypatch):
    """The whole sampling forward under datapath bf16: plane-fed (norms and output stages write planes, GEMMs read the hi planes) == fp32-fed."""
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    unet = UNet2DCondition(UNetConfig.named(family), DEV)
    unet.params.init_synthetic(2)
    unet.params.pack_bf16(bwd=False)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 4, 16, 16, generator=g).to(DEV)
    t = torch.tensor([981, 21, 501, 481], dtype=torch.int32, device=DEV)
    c = torch.randn(4, 77, ctx_dim, generator=g).to(DEV)
    out = {}
    for on in (False, True):
        monkeypatch.setattr(L, "BF16_PLANES", on)
        out[on] = unet(x, t, c).clone()
    assert torch.equal(out[False], out[True])
