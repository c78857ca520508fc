"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle / a plain torch fp32 reference."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ddpo_amd import lib as L
from oracle import prng as OP, ppo as OPPO
from oracle.ddim import DDIMOracle
from oracle.optim import AdamWBf16Mu, AccumulatingState

DEV = "cuda"


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ------------------------------------------------------------------------------------------------ PRNG
@pytest.mark.parametrize("n", [1, 2, 7, 4096, 2 * 4 * 8 * 8, 8 * 4 * 64 * 64, 8 * 4 * 64 * 64 + 3])
def test_threefry_bits_bit_exact(n):
    key = OP.PRNGKey(1234)
    out, bits = L.threefry_normal(key, (n,), return_bits=True)
    want_bits = OP.random_bits(key, n)
    got_bits = bits.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_bits, want_bits)                      # integer work: bit-exact
    assert np.array_equal(L.threefry_bits_host(key, n), want_bits)
    want = OP.normal(key, (n,))
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-6, atol=2e-6)   # fp32 erfinv: few ulp


def test_threefry_matches_jax_doc_values():
    for seed, shape, want in ((0, (1,), [-0.20584226]), (42, (), [-0.18471177]), (0, (3,), [1.8160863, -0.48262316, 0.33988908])):
        got = L.threefry_normal(OP.PRNGKey(seed), shape if shape else (1,)).cpu().numpy().ravel()
        np.testing.assert_allclose(got, np.asarray(want, dtype=np.float32), rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ DDIM step
def _sched(pred="epsilon", T=50):
    from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False,
                      steps_offset=1, prediction_type=pred)
    st = s.set_timesteps(s.create_state(device=DEV), T)
    d = DDIMOracle(prediction_type=pred)
    ost = d.set_timesteps(d.create_state(), T)
    return s, st, d, ost


def test_scheduler_integer_state_bit_exact():
    s, st, d, ost = _sched()
    assert np.array_equal(st.timesteps, ost.timesteps) and st.timesteps.dtype == np.int32
    assert np.array_equal(st.common.alphas_cumprod, ost.alphas_cumprod)
    assert np.array_equal(st.common.alphas_cumprod_dev.cpu().numpy(), ost.alphas_cumprod)
    s4 = s.set_timesteps(st, 4)
    assert s4.timesteps.tolist() == [751, 501, 251, 1]


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("shape", [(2, 4, 8, 8), (8, 4, 64, 64), (3, 4, 96, 96)])
def test_ddim_step_fwd(pred, shape):
    s, st, d, ost = _sched(pred)
    rng = np.random.default_rng(0)
    eu, ec, x, z = (rng.standard_normal(shape, dtype=np.float32) for _ in range(4))
    B = shape[0]
    ts = np.asarray([981, 1, 481, 21, 961, 501, 41, 241][:B], dtype=np.int32)
    consts = s.kernel_consts(st, 1.0)
    xn, lp = L.ddim_step_fwd(*(torch.from_numpy(a).to(DEV) for a in (eu, ec, x, z)), torch.from_numpy(ts).to(DEV), 5.0, consts)
    guided = (eu + np.float32(5.0) * (ec - eu)).astype(np.float32)
    oxn, olp = d.step(ost, guided, ts, x, noise=z, eta=1.0)
    assert _rel(xn.cpu().numpy(), oxn) < 1e-5
    np.testing.assert_allclose(lp.cpu().numpy(), olp, rtol=1e-4, atol=1e-4)


def test_scheduler_step_mirror_modes():
    s, st, d, ost = _sched()
    rng = np.random.default_rng(1)
    shape = (2, 4, 8, 8)
    e, x = (rng.standard_normal(shape, dtype=np.float32) for _ in range(2))
    key = OP.PRNGKey(5)
    prev, _, lp = s.step(st, torch.from_numpy(e).to(DEV), 501, torch.from_numpy(x).to(DEV), key=key, eta=1.0)
    oprev, olp = d.step(ost, e, 501, x, noise=OP.normal(key, shape), eta=1.0)
    assert _rel(prev.cpu().numpy(), oprev) < 1e-5
    np.testing.assert_allclose(lp.cpu().numpy(), olp, rtol=1e-4, atol=1e-4)
    _, _, lp2 = s.step(st, torch.from_numpy(e).to(DEV), np.array([501, 501]), torch.from_numpy(x).to(DEV), prev_sample=prev, eta=1.0)
    np.testing.assert_allclose(lp2.cpu().numpy(), olp, rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        s.step(st, torch.from_numpy(e).to(DEV), 501, torch.from_numpy(x).to(DEV), key=key, prev_sample=prev, eta=1.0)


# ------------------------------------------------------------------------------------------------ PPO fwd/bwd
@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("train_cfg", [True, False])
def test_ppo_fwd_bwd(pred, train_cfg):
    s, st, d, ost = _sched(pred)
    rng = np.random.default_rng(2)
    B, shape = 6, (6, 4, 16, 16)
    ec, eu, x, z = (rng.standard_normal(shape, dtype=np.float32) for _ in range(4))
    ts = np.asarray([981, 481, 1, 21, 701, 241], dtype=np.int32)
    guided = (eu + np.float32(5.0) * (ec - eu)).astype(np.float32) if train_cfg else ec
    xn, lp0 = d.step(ost, guided, ts, x, noise=z, eta=1.0)
    old = (lp0 + np.asarray([0.0, 5e-5, -3e-4, 3e-4, 3e-4, -3e-4], dtype=np.float32)).astype(np.float32)
    adv = np.asarray([1.5, -0.7, 12.0, -20.0, 0.3, -0.2], dtype=np.float32)
    consts = s.kernel_consts(st, 1.0)
    t = lambda a: torch.from_numpy(a).to(DEV)
    d_c, d_u, per, info = L.ddim_logprob_ppo_fwd_bwd(t(ec), t(eu) if train_cfg else None, t(x), t(xn), t(ts), t(old), t(adv),
                                                      5.0, 1e-4, train_cfg, consts)
    loss, oinfo, olp, odc, odu = OPPO.closed_form_numpy(d, ost, ec, eu, x, xn, ts, old, adv, 5.0, 1.0, 1e-4, train_cfg)
    np.testing.assert_allclose(per[:, 0].cpu().numpy(), olp, rtol=1e-4, atol=1e-4)
    info = info.cpu().numpy()
    assert info[2] == pytest.approx(float(loss), rel=1e-4, abs=1e-5)
    assert info[1] == pytest.approx(float(oinfo["clipfrac"]), abs=1e-6)
    assert info[0] == pytest.approx(float(oinfo["approx_kl"]), rel=1e-2, abs=1e-9)
    scale = np.abs(odc).max()
    np.testing.assert_allclose(d_c.cpu().numpy(), odc, rtol=2e-3, atol=2e-4 * scale)
    if train_cfg:
        np.testing.assert_allclose(d_u.cpu().numpy(), odu, rtol=2e-3, atol=2e-4 * scale)


# ------------------------------------------------------------------------------------------------ optimizer
@pytest.mark.parametrize("mu_bf16", [True, False])
def test_adamw_matches_oracle(mu_bf16):
    rng = np.random.default_rng(3)
    n = 100003 * 4
    p0 = rng.standard_normal(n).astype(np.float32)
    opt = AdamWBf16Mu(mu_decay_in_bf16=mu_bf16)
    ost = AccumulatingState([p0], opt)
    p = torch.from_numpy(p0.copy()).to(DEV)
    g = torch.zeros(n, device=DEV)
    mu = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    nu = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        scale = [3.0, 1e-3, 0.5][step - 1]            # first step clips, second does not
        g1 = (rng.standard_normal(n) * scale).astype(np.float32)
        g2 = (rng.standard_normal(n) * scale).astype(np.float32)
        ost.apply_gradients([g1], False)
        ost.apply_gradients([g2], True)
        g += torch.from_numpy(g1).to(DEV)
        g += torch.from_numpy(g2).to(DEV)
        sq = L.grad_sqnorm(g)
        norm = math.sqrt(float(sq.item())) * 0.5
        assert norm == pytest.approx(float(ost.last_grad_norm), rel=1e-5)
        L.adamw_bf16mu_step(p, g, mu, nu, sq, 0.5, 1e-5, 0.9, 0.999, 1e-8, 1e-4, 1.0, step, mu_decay_in_bf16=mu_bf16)
        assert float(g.abs().max()) == 0.0
        np.testing.assert_allclose(p.cpu().numpy(), ost.params[0], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(nu.cpu().numpy(), ost.opt_state["nu"][0], rtol=1e-5, atol=1e-12)
        mism = (mu.float().cpu().numpy() != ost.opt_state["mu"][0]).mean()
        assert mism < 1e-3      # bf16 first moment identical except rare 1-ulp rounding ties


# ------------------------------------------------------------------------------------------------ norms / elementwise
@pytest.mark.parametrize("B,HW,C", [(2, 64, 32), (2, 64, 96), (3, 256, 320), (2, 1024, 640), (2, 64, 1280), (2, 64, 1920), (1, 64, 2560), (2, 4096, 128),
                                    (2, 256, 2560), (2, 1024, 1920), (2, 900, 1280), (1, 1024, 2560)])      # round 5: the one-launch form (HW <= 256) at its widest group, and 32x32-level shapes on the three-launch path
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm(B, HW, C, silu):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * HW, C, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = L.groupnorm(x.to(DEV), B, HW, gamma.to(DEV), beta.to(DEV), 32, 1e-5, silu)
    ref = torch.nn.functional.group_norm(x.view(B, HW, C).permute(0, 2, 1).double(), 32, gamma.double(), beta.double(), 1e-5)
    if silu:
        ref = torch.nn.functional.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(B * HW, C)
    assert _rel(y.cpu().numpy(), ref.numpy()) < 2e-5


def test_groupnorm_workspace_reuse_is_bit_reproducible():
    """One workspace is reused by back-to-back GroupNorm launches of different geometries (chunk counts 1 .. 125, B up to 1030) in a tight
    loop with nothing waiting in between: every result must equal the float64 reference and a repeated call must be BIT-identical
    (fixed-order reductions).  (Round 3 tried finishing the statistics in the last-arriving block of the statistics kernel — arrival
    counters, write-through partials — to save the finalize launch: correct under this test, 2.7 % SLOWER on the sampling bench,
    profiles/r03_ab_gn_ticket.log; not kept.)"""
    g = torch.Generator().manual_seed(3)
    geoms = [(3, 4096, 320), (40, 64, 64), (1, 25, 96), (16, 1024, 640), (2, 4096, 32), (5, 9, 1280), (1030, 4, 32)]
    data = []
    for B, HW, C in geoms:
        x = (torch.randn(B * HW, C, generator=g) * 2 + 0.5).to(DEV)
        ga, be = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        data.append((B, HW, C, x, ga, be))
    first = None
    for rep in range(6):
        outs = [L.groupnorm(x, B, HW, ga, be, 32, 1e-5, False) for B, HW, C, x, ga, be in data]      # queued back to back on one stream
        if first is None:
            first = outs
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, outs))
    for (B, HW, C, x, ga, be), y in zip(data, first):
        ref = torch.nn.functional.group_norm(x.cpu().view(B, HW, C).permute(0, 2, 1).double(), 32, ga.cpu().double(), be.cpu().double(), 1e-5)
        assert _rel(y.cpu().numpy(), ref.permute(0, 2, 1).reshape(B * HW, C).numpy()) < 2e-5


@pytest.mark.parametrize("rows,C", [(64, 32), (77, 64), (256, 320), (100, 640), (64, 1280)])
def test_layernorm(rows, C):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, C, generator=g) * 3 + 1
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = L.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5)
    ref = torch.nn.functional.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5)
    assert _rel(y.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("rows,C", [(4096, 320), (16384 + 3, 640), (5000, 64), (4097, 512)])
def test_layernorm_batched_rows_kernel_is_bit_identical_per_row(rows, C, monkeypatch):
    """Round 4: rows >= 4096 of C <= 512 channels run layernorm_rows_kernel (four rows requested per wave before the first is reduced); a row's
    arithmetic is the one-row-per-wave kernel's — the training forward of a sub-batch must see the sampler's bits — for the fp32 result and for
    both plane formats."""
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3 + 1).to(DEV)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    y = L.layernorm(x, gamma, beta, 1e-5)
    ref = torch.nn.functional.layer_norm(x.cpu().double(), (C,), gamma.cpu().double(), beta.cpu().double(), 1e-5)
    assert _rel(y.cpu().numpy(), ref.numpy()) < 1e-5
    sub = [slice(0, 100), slice(rows - 1001, rows)]              # < 4096 rows: the one-row-per-wave kernel
    for sl in sub:
        assert torch.equal(L.layernorm(x[sl].contiguous(), gamma, beta, 1e-5), y[sl])
    monkeypatch.setattr(L, "DATAPATH", "bf16x3")
    for fmt in ((1, 2) if C % 32 == 0 else (1,)):
        pl = L.layernorm(x, gamma, beta, 1e-5, planes=fmt)
        for sl in sub:
            ps = L.layernorm(x[sl].contiguous(), gamma, beta, 1e-5, planes=fmt)
            assert torch.equal(ps.plane("hi"), pl.plane("hi")[sl]) and torch.equal(ps.plane("lo"), pl.plane("lo")[sl])
        rf = L.split_planes(y, fmt=fmt - 1)
        assert torch.equal(pl.plane("hi"), rf.plane("hi")) and torch.equal(pl.plane("lo"), rf.plane("lo"))


def test_small_elementwise():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(50, 2 * 128, generator=g) * 2
    y = L.geglu(x.to(DEV)).cpu()
    ref = x[:, :128] * torch.nn.functional.gelu(x[:, 128:].double(), approximate="tanh")
    assert _rel(y.numpy(), ref.numpy()) < 1e-5
    assert _rel(L.silu(x.to(DEV)).cpu().numpy(), torch.nn.functional.silu(x.double()).numpy()) < 1e-5
    ts = torch.tensor([981, 1, 500], dtype=torch.int32)
    emb = L.timestep_embedding(ts.to(DEV), 320).cpu()
    from oracle.unet import timestep_embedding
    np.testing.assert_allclose(emb.numpy(), timestep_embedding(ts, 320).numpy(), rtol=0, atol=2e-4)
    a = torch.randn(2, 4, 8, 8, generator=g)
    nhwc = L.nchw_to_nhwc(a.to(DEV))
    assert torch.equal(nhwc.cpu().view(2, 8, 8, 4), a.permute(0, 2, 3, 1))
    assert torch.equal(L.nhwc_to_nchw(nhwc, 2, 4, 8, 8).cpu(), a)
    dst = torch.zeros(10, 24, device=DEV)
    src = torch.randn(10, 8, generator=g)
    L.copy_cols(src.to(DEV), dst, 12, 10, 8)
    assert torch.equal(dst.cpu()[:, 12:20], src) and float(dst.cpu()[:, :12].abs().sum()) == 0
    s = torch.randn(33, 100, generator=g)
    sm = L.softmax_rows_(s.clone().to(DEV), 0.7).cpu()
    assert _rel(sm.numpy(), torch.softmax(s.double() * 0.7, -1).numpy()) < 1e-5
    c = L.scale_shift_clip(s.to(DEV), 0.5, 0.5, 0.0, 1.0).cpu()
    assert torch.allclose(c, (s / 2 + 0.5).clamp(0, 1))


# ------------------------------------------------------------------------------------------------ GEMM / conv
@pytest.mark.parametrize("M,K,N", [(64, 32, 64), (77 * 2, 64, 128), (1000, 320, 320), (4096, 1280, 640), (130, 36, 4), (16, 1280, 320),
                                   (8192, 320, 2560), (256, 5120, 1280)])
def test_gemm_dense(M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    out = L.linear(a.to(DEV), w.to(DEV), bias.to(DEV), residual=res.to(DEV)).cpu()
    ref = a.double() @ w.double() + bias.double() + res.double()
    assert _rel(out.numpy(), ref.numpy()) < 2e-6 * math.sqrt(K) + 1e-6
    # transposed-weight form and alpha
    out2 = L.gemm_conv(a.to(DEV), w.t().contiguous().to(DEV), M=M, N=N, K=K, w_trans=True, alpha=0.25).cpu() if K % 4 == 0 else None
    if out2 is not None:
        assert _rel(out2.numpy(), (0.25 * (a.double() @ w.double())).numpy()) < 2e-6 * math.sqrt(K) + 1e-6


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride,ups", [
    (2, 8, 8, 4, 32, 3, 1, False), (2, 8, 8, 32, 4, 3, 1, False), (2, 8, 8, 96, 64, 3, 1, False), (2, 8, 8, 64, 64, 3, 2, False),
    (2, 4, 4, 128, 128, 3, 1, True), (2, 8, 8, 64, 128, 1, 1, False), (1, 64, 64, 320, 320, 3, 1, False), (2, 32, 32, 640, 640, 3, 2, False),
    (1, 16, 16, 1280, 1280, 3, 1, True), (3, 5, 7, 32, 64, 3, 1, False), (3, 5, 7, 32, 64, 3, 2, False)])
def test_conv2d(B, H, W, Cin, Cout, ks, stride, ups):
    g = torch.Generator().manual_seed(H + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(ks, ks, Cin, Cout, generator=g) / math.sqrt(ks * ks * Cin)
    bias = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout, generator=g)
    xin = x.permute(0, 3, 1, 2).double()
    if ups:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xin, w.permute(3, 2, 0, 1).double(), bias.double(), stride=stride, padding=ks // 2)
    OH, OW = ref.shape[2], ref.shape[3]
    ref = ref + temb.double()[:, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(B * OH * OW, Cout)
    res = torch.randn(B * OH * OW, Cout, generator=g)
    out, oh, ow = L.conv2d(x.reshape(-1, Cin).to(DEV), w.to(DEV), bias.to(DEV), B, H, W, Cin, Cout, ks, stride=stride, upsample=ups,
                           rowbias=temb.to(DEV), rows_per_batch=OH * OW, residual=res.to(DEV))
    assert (oh, ow) == (OH, OW)
    assert _rel(out.cpu().numpy(), (ref + res.double()).numpy()) < 2e-6 * math.sqrt(ks * ks * Cin) + 1e-6


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,heads,Nq,Nk,d", [(2, 8, 64, 64, 4), (2, 8, 64, 77, 8), (1, 8, 256, 256, 16), (2, 8, 1024, 1024, 40),
                                             (2, 8, 1024, 77, 40), (1, 8, 256, 256, 80), (2, 8, 64, 64, 160), (2, 8, 64, 77, 160),
                                             (1, 5, 200, 333, 64), (1, 8, 4096, 4096, 40)])
def test_attention(B, heads, Nq, Nk, d):
    g = torch.Generator().manual_seed(Nq + Nk + d)
    C = heads * d
    q = torch.randn(B * Nq, C, generator=g)
    k = torch.randn(B * Nk, C, generator=g)
    v = torch.randn(B * Nk, C, generator=g)
    out = L.attention(q.to(DEV), k.to(DEV), v.to(DEV), B, heads, Nq, Nk, d).cpu()
    sp = lambda t, n: t.view(B, n, heads, d).permute(0, 2, 1, 3).double()
    s = torch.softmax(sp(q, Nq) @ sp(k, Nk).transpose(-1, -2) * d ** -0.5, -1)
    ref = (s @ sp(v, Nk)).permute(0, 2, 1, 3).reshape(B * Nq, C)
    assert _rel(out.numpy(), ref.numpy()) < 1e-5


def test_attention_large_logits_online_softmax():
    """Forces the running-max rescale: one key tile carries a spike that arrives late in the sweep."""
    B, heads, N, d = 1, 8, 256, 40
    g = torch.Generator().manual_seed(0)
    q = torch.randn(N, heads * d, generator=g)
    k = torch.randn(N, heads * d, generator=g)
    v = torch.randn(N, heads * d, generator=g)
    k[200] = q[5] * 4.0
    out = L.attention(q.to(DEV), k.to(DEV), v.to(DEV), B, heads, N, N, d).cpu()
    sp = lambda t: t.view(B, N, heads, d).permute(0, 2, 1, 3).double()
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * d ** -0.5, -1) @ sp(v)).permute(0, 2, 1, 3).reshape(N, heads * d)
    assert _rel(out.numpy(), ref.numpy()) < 1e-5


def test_bad_arguments_fail_loudly():
    x = torch.zeros(8, 6, device=DEV)
    with pytest.raises(L.DdpoHipError):
        L.layernorm(x, torch.ones(6, device=DEV), torch.zeros(6, device=DEV))       # C % 4 != 0
    with pytest.raises(L.DdpoHipError):
        L.attention(x, x, x, 1, 2, 8, 8, 3)                                          # unsupported head dim
