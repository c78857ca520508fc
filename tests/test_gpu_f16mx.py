"""f16mx forward operator (ABI v7: ddpo_pack_weights_f16mx / ddpo_split_planes_f16mx / ddpo_gemm_conv_fwd_f16mx_planes) against a host
emulation built from torch's float8 / float16 conversions: (1) the plane encodings bit for bit, (2) the GEMM against the exact
product of the DECODED planes (layouts, lane pairing, block scales: only fp32 accumulation error left), (3) the datapath's accuracy
against float64 next to bf16x3.  The operator is not routed by the models (ddpo_amd/lib.py); tools/native/kernel_probe mx times it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from ddpo_amd import lib as L

pytestmark = pytest.mark.gpu


def _dec_planes(p16, p8):
    """(h, h8, l8) as float64 (rows, C) from the activation planes."""
    rows, C = p16.shape
    h = p16.view(torch.float16).double()
    b = p8.view(torch.float8_e5m2).double().view(rows, C // 32, 2, 2, 16)      # per block: [k half][h8 | l8][16]
    return h, b[:, :, :, 0].reshape(rows, C), b[:, :, :, 1].reshape(rows, C) / 2048.0


def _dec_weights(wp):
    """(h, h8, l8) as float64 (K, N) from the weight planes (zero padded rows dropped)."""
    K, N = wp["K"], wp["N"]
    s = torch.pow(2.0, wp["scale"].double() - 127.0)                      # (N,)
    h = wp["w16"].view(torch.float16).double().permute(0, 2, 1).reshape(-1, N)[:K]      # (Kb, N, 32) -> (Kb*32, N)
    b = wp["w8"].view(torch.float8_e4m3fn).double().view(-1, N, 2, 2, 16)      # (Kb, N, k half, [l8 | h8], 16)
    l8 = (b[:, :, :, 0].reshape(-1, N, 32) * s[None, :, None] / 2048.0).permute(0, 2, 1).reshape(-1, N)[:K]
    h8 = (b[:, :, :, 1].reshape(-1, N, 32) * s[None, :, None]).permute(0, 2, 1).reshape(-1, N)[:K]
    return h, h8, l8


def test_activation_planes_are_f16_and_e5m2_of_the_split():
    torch.manual_seed(0)
    x = (torch.randn(77, 96, device="cuda") * torch.logspace(-6, 3, 96, device="cuda")).contiguous()
    p16, p8 = L.split_planes_f16mx(x)
    h = x.half()
    assert torch.equal(p16.view(torch.float16), h)                       # round to nearest even, like torch
    l = (x - h.float()) * 2048.0
    b = p8.view(torch.float8_e5m2).view(77, 3, 2, 2, 16)
    assert torch.equal(b[:, :, :, 0].reshape(77, 96).float(), h.float().to(torch.float8_e5m2).float())
    assert torch.equal(b[:, :, :, 1].reshape(77, 96).float(), l.to(torch.float8_e5m2).float())
    hh, h8, l8 = _dec_planes(p16, p8)
    xd = x.double()
    assert float(((hh + l8) - xd).abs().max() / xd.abs().max()) < 2.0 ** -13          # h + l8: 11 + 3 bits
    big = xd.abs() >= 2.0 ** -14                                          # e5m2's normal range (smaller values lose bits, then flush)
    assert float((h8 - xd).abs().div(xd.abs())[big].max()) < 2.0 ** -2.5  # e5m2 of h: 3 significant bits


def test_weight_planes_are_f16_and_column_scaled_e4m3():
    torch.manual_seed(1)
    K, N = 100, 72                                                        # ragged K (zero padded to 128), N % 32 != 0
    w = (torch.randn(K, N, device="cuda") * torch.logspace(-3, 1, N, device="cuda")[None, :]).contiguous()
    w[:, 5] = 0.0                                                         # an all-zero column keeps a valid scale
    wp = L.pack_weights_f16mx(w)
    h, h8, l8 = _dec_weights(wp)
    assert torch.equal(h.float(), w.half().float())
    e = torch.floor(torch.log2(w.abs().amax(0).clamp_min(2.0 ** -95))) - 7            # scale exponent: max|w| / 2^e in [128, 256)
    assert torch.equal(wp["scale"].double()[w.abs().amax(0) > 0] - 127.0, e.double()[w.abs().amax(0) > 0])
    s = torch.pow(2.0, wp["scale"].float() - 127.0)
    q = lambda v: (v / s).to(torch.float8_e4m3fn).float() * s
    assert torch.equal(h8.float(), q(w.half().float()))
    assert torch.equal((l8 * 2048.0).float(), q((w - w.half().float()) * 2048.0))
    assert float(wp["w16"][3, :, 4:].abs().max()) == 0 and float(wp["w8"].view(-1, N, 2, 2, 16)[3, :, 0, :, 4:].abs().max()) == 0      # k >= K: zeros


CASES = [  # (B, H, Cin, Cout, ksize, stride, upsample)  ksize 0: dense with M = B * H rows
    (2, 16, 64, 96, 3, 1, 0),        # 128x64 tiles, ragged N tile
    (2, 32, 320, 320, 3, 1, 0),      # 128x320 tiles
    (2, 16, 640, 640, 3, 1, 0),      # split-K
    (2, 16, 128, 128, 3, 2, 0),      # stride 2
    (2, 8, 64, 64, 3, 1, 1),         # nearest-neighbour upsample in the gather
    (2, 16, 96, 160, 1, 1, 0),       # 1x1
    (3, 77, 768, 320, 0, 1, 0),      # dense, ragged M
    (1, 640, 320, 1280, 0, 1, 0),    # dense, 128x128 tiles
]


@pytest.mark.parametrize("B,H,Cin,Cout,ks,stride,ups", CASES)
def test_gemm_equals_the_product_of_the_decoded_planes(B, H, Cin, Cout, ks, stride, ups):
    torch.manual_seed(2)
    conv = ks > 0
    rows = B * H * H if conv else B * H
    K = ks * ks * Cin if conv else Cin
    x = torch.randn(rows, Cin, device="cuda")
    w = torch.randn(K, Cout, device="cuda") / K ** 0.5
    bias = torch.randn(Cout, device="cuda")
    planes, wp = L.split_planes_f16mx(x), L.pack_weights_f16mx(w)
    ah, ah8, al8 = _dec_planes(*planes)
    wh, wh8, wl8 = _dec_weights(wp)
    if conv:
        pad = ks // 2
        VH = 2 * H if ups else H
        OH = (VH + 2 * pad - ks) // stride + 1
        M = B * OH * OH
        geom = dict(ksize=ks, stride=stride, pad=pad, upsample=ups, B=B, H=H, W=H, Cin=Cin, OH=OH, OW=OH)

        def cv(a, b_):
            a = a.view(B, H, H, Cin).permute(0, 3, 1, 2)
            if ups:
                a = a.repeat_interleave(2, 2).repeat_interleave(2, 3)
            return TF.conv2d(a, b_.view(ks, ks, Cin, Cout).permute(3, 2, 0, 1), None, stride=stride, padding=pad).permute(0, 2, 3, 1).reshape(M, Cout)
    else:
        M, geom = rows, None
        cv = lambda a, b_: a @ b_
    res = torch.randn(M, Cout, device="cuda")
    out = L.gemm_conv_f16mx(planes, wp, M=M, bias=bias, residual=res, conv=geom)
    ref_planes = cv(ah, wh) + cv(ah8, wl8) + cv(al8, wh8) + bias.double() + res.double()
    ref_true = cv(x.double(), w.double()) + bias.double() + res.double()
    scale = float(ref_true.abs().max())
    assert float((out.double() - ref_planes).abs().max()) < 3e-6 * scale            # the operator's contract (fp32 accumulation only)
    err = float((out.double() - ref_true).pow(2).mean().sqrt() / (ref_true - bias.double() - res.double()).pow(2).mean().sqrt())
    assert err < 3e-5                                                     # the datapath: ~1e-5 rms on Gaussian operands (bf16x3: ~3e-6)


def test_output_stage_emits_f16mx_planes_of_the_result():
    torch.manual_seed(3)
    M, K, N = 300, 256, 320
    x, w = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda") / 16
    out, (p16, p8) = L.gemm_conv_f16mx(L.split_planes_f16mx(x), L.pack_weights_f16mx(w), M=M, planes_out=True)
    q16, q8 = L.split_planes_f16mx(out)
    assert torch.equal(p16, q16) and torch.equal(p8, q8)


def test_operands_beyond_the_f16_range_saturate_instead_of_overflowing():
    """every f16 split of the f16mx datapath (activation planes: common.h mx_split4; the attention's V / dO / K / Q images:
    split2h in attention_bf16.hip / attention_bwd_bf16.hip) clamps to +-65504 (60000 in the backward) before the conversion, so an
    activation beyond the f16 range degrades to a saturated value — never to inf, and never to the NaN an inf - inf low part would be.
    (Round 5: the first run of this test found two holes in mx_split4 — the low part was taken from the UNCLAMPED value, and f16 values above
    61440 round to e5m2 infinity in the 8-bit image of h; both parts now derive from the clamped value, h8 from h clamped to 57344.)"""
    old = L.DATAPATH
    L.DATAPATH = "f16mx"
    try:
        x = torch.tensor([[7.0e4, -3.0e5, 65504.0, 1.0e30] * 8], device="cuda").repeat(16, 1).contiguous()       # (16, 32)
        p16, p8 = L.split_planes_f16mx(x)
        h = p16.view(torch.float16).float()
        assert torch.isfinite(h).all() and float(h.abs().max()) == 65504.0
        assert torch.isfinite(p8.view(torch.float8_e5m2).float()).all()
        torch.manual_seed(3)
        B, heads, Nq, Nk, d = 1, 2, 64, 96, 40
        q = torch.randn(B * Nq, heads * d, device="cuda")
        k = torch.randn(B * Nk, heads * d, device="cuda")
        v = torch.randn(B * Nk, heads * d, device="cuda")
        v[::7] *= 1.0e6                                                   # far beyond the f16 range
        out = L.attention(q, k, v, B, heads, Nq, Nk, d)
        assert torch.isfinite(out).all()
        ref = torch.softmax((q.view(Nq, heads, d).transpose(0, 1) @ k.view(Nk, heads, d).transpose(0, 1).transpose(1, 2)) * d ** -0.5, -1) @ \
            v.clamp(-65504.0, 65504.0).view(Nk, heads, d).transpose(0, 1)
        ref = ref.transpose(0, 1).reshape(Nq, heads * d)
        assert float((out - ref).abs().max() / ref.abs().max()) < 1e-3   # = the attention of the SATURATED values
    finally:
        L.DATAPATH = old


@pytest.mark.parametrize("B,H,Cin,Cout,ks,ups", [(16, 64, 320, 320, 3, 0), (16, 64, 960, 320, 3, 0), (16, 32, 640, 640, 3, 1), (13, 64, 320, 320, 3, 0),
                                                 (1, 65500, 2560, 320, 0, 0)])
def test_tall_tile_is_bit_identical_to_the_128_row_tiles(B, H, Cin, Cout, ks, ups, monkeypatch):
    """the f16mx 256 x 320 tile (APL = 7: eight waves of 64 x 160, two-phase k-tile) is routed by default where its grid fills the
    chip and keeps the per-accumulator order of the 128-row kernel (f16 k-half 0, f16 k-half 1, MX).  DDPO_MX_TALL=0 (read per launch)
    keeps the layer on the 128-row tiles: the two outputs must be equal bit for bit over the whole tensor, the tall-tile counter must have
    moved only in the default run; 208 tall tiles (B = 13) and a ragged last row tile (dense, M = 65500) must agree too."""
    torch.manual_seed(7)
    conv = ks > 0
    rows, K = (B * H * H, 9 * Cin) if conv else (B * H, Cin)
    VH = 2 * H if ups else H
    M = B * VH * VH if conv else rows
    x = torch.randn(rows, Cin, device="cuda")
    w = torch.randn(K, Cout, device="cuda") / K ** 0.5
    bias, res = torch.randn(Cout, device="cuda"), torch.randn(M, Cout, device="cuda")
    planes, wp = L.split_planes_f16mx(x), L.pack_weights_f16mx(w)
    geom = dict(ksize=3, stride=1, pad=1, upsample=ups, B=B, H=H, W=H, Cin=Cin, OH=VH, OW=VH) if conv else None
    outs, tall = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DDPO_MX_TALL", mode)
        before = L.gemm_tile_launch_counts()["tall_256x320"]
        outs[mode] = L.gemm_conv_f16mx(planes, wp, M=M, bias=bias, residual=res, conv=geom).clone()
        torch.cuda.synchronize()
        tall[mode] = L.gemm_tile_launch_counts()["tall_256x320"] - before
    assert tall["0"] == 0 and tall["1"] == 1, tall
    assert torch.equal(outs["0"], outs["1"])


@pytest.mark.parametrize("B,H,Cin,Cout,ks", [(2, 32, 320, 320, 3), (16, 64, 320, 320, 3), (2, 16, 640, 640, 3), (1, 640, 320, 1280, 0)])
def test_single_pass_f16_is_the_f16_term_of_the_operator(B, H, Cin, Cout, ks):
    """ABI v14, opt-in (DDPO_MX_CROSS=0): both 8-bit planes NULL at ddpo_gemm_conv_fwd_f16mx_planes runs a_h * w_h alone on the single-plane kernels
    (128-row tiles on the three-weight-stage loop, the 256 x 320 tile on the four-stage ring).  Contract: the exact product of the DECODED f16
    planes up to fp32 accumulation, i.e. the f16mx result minus its two cross terms; ~3e-4 rms against float64 on Gaussian operands."""
    from ctypes import byref
    torch.manual_seed(5)
    conv = ks > 0
    rows, K = (B * H * H, ks * ks * Cin) if conv else (B * H, Cin)
    M = rows
    x = torch.randn(rows, Cin, device="cuda")
    w = torch.randn(K, Cout, device="cuda") / K ** 0.5
    (p16, p8), wp = L.split_planes_f16mx(x), L.pack_weights_f16mx(w)
    ah, _, _ = _dec_planes(p16, p8)
    wh, _, _ = _dec_weights(wp)
    if conv:
        cv = lambda a, b_: TF.conv2d(a.view(B, H, H, Cin).permute(0, 3, 1, 2), b_.view(ks, ks, Cin, Cout).permute(3, 2, 0, 1), None, padding=ks // 2).permute(0, 2, 3, 1).reshape(M, Cout)
    else:
        cv = lambda a, b_: a @ b_
    d = L.GemmDesc()
    out = torch.empty(M, Cout, device="cuda")
    d.out = out.data_ptr(); d.ld_out = Cout; d.alpha = 1.0; d.M, d.N, d.K = M, Cout, K; d.w_layout = 1
    if conv:
        for k_, v_ in dict(ksize=ks, stride=1, pad=ks // 2, upsample=0, B=B, H=H, W=H, Cin=Cin, OH=H, OW=H).items():
            setattr(d, k_, v_)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    f = L.load().ddpo_gemm_conv_fwd_f16mx_planes
    assert f(byref(d), L._p(p16), None, Cin, L._p(wp["w16"]), L._p(wp["w8"]), L._p(ws), ws.numel(), None) == -1          # exactly one NULL: refused
    assert f(byref(d), L._p(p16), None, Cin, L._p(wp["w16"]), None, L._p(ws), ws.numel(), None) == 0
    torch.cuda.synchronize()
    ref_planes, ref_true = cv(ah, wh), cv(x.double(), w.double())
    scale = float(ref_true.abs().max())
    assert float((out.double() - ref_planes).abs().max()) < 3e-6 * scale
    err = float((out.double() - ref_true).pow(2).mean().sqrt() / ref_true.pow(2).mean().sqrt())
    assert 5e-5 < err < 6e-4, err
