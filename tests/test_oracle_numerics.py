"""CPU checks of the oracle restatements: DDIM schedule anchors, log-prob identities, PPO closed form vs autograd,
optax-AdamW restatement vs an independent float64 derivation, parameter-count anchors."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import ppo, prng
from oracle.ddim import DDIMOracle
from oracle.optim import AdamWBf16Mu, AccumulatingState, bf16_round, global_norm
from oracle import unet as OU


def _gold(golden_dir):
    with open(os.path.join(golden_dir, "ddim_schedule.json")) as f:
        return json.load(f)


def test_schedule_anchors(golden_dir):
    g = _gold(golden_dir)
    d = DDIMOracle()
    s = d.create_state()
    for k, v in g["alphas_cumprod"].items():
        assert abs(float(s.alphas_cumprod[int(k)]) - v) <= 2e-7 * max(1.0, abs(v)) + 5e-9, k
    assert float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])
    s50 = d.set_timesteps(s, 50)
    assert s50.timesteps[:3].tolist() == g["timesteps_T50_head"] and s50.timesteps[-3:].tolist() == g["timesteps_T50_tail"]
    assert d.set_timesteps(s, 4).timesteps.tolist() == g["timesteps_T4"]
    for c in g["coeffs_eta1"]:
        st = d.set_timesteps(s, c["T"])
        a_t, a_p, b_t, std = d.coefficients(st, c["t"], 1.0)
        dmu = np.sqrt(1 - a_p - std ** 2) - np.sqrt(a_p) * np.sqrt(1 - a_t) / np.sqrt(a_t)
        assert abs(float(std) - c["sigma"]) < 2e-6 and abs(float(dmu) - c["dmu_deps"]) < 5e-6


def test_param_count_anchors(golden_dir):
    g = _gold(golden_dir)["param_counts"]
    assert OU.count_params(OU.unet_param_shapes(OU.SD15)) == g["unet_sd15"]
    assert OU.count_params(OU.unet_param_shapes(OU.SD21)) == g["unet_sd21"]
    assert OU.count_params(OU.vae_decoder_param_shapes(OU.VAE_SD)) == g["vae_decoder_with_post_quant"]


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
def test_sampling_logprob_identity(pred):
    """In sampling mode x' - mu = sigma z, so log_prob = mean(-z^2/2) - log sigma - log sqrt(2 pi)."""
    d = DDIMOracle(prediction_type=pred)
    st = d.set_timesteps(d.create_state(), 50)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 4, 8, 8), dtype=np.float32)
    e = rng.standard_normal((3, 4, 8, 8), dtype=np.float32)
    z = rng.standard_normal((3, 4, 8, 8), dtype=np.float32)
    t = 481
    xn, lp = d.step(st, e, t, x, noise=z, eta=1.0)
    _, _, _, std = d.coefficients(st, t, 1.0)
    want = (-(z.astype(np.float64) ** 2) / 2).reshape(3, -1).mean(1) - math.log(float(std)) - 0.5 * math.log(2 * math.pi)
    np.testing.assert_allclose(lp, want, rtol=0, atol=2e-4)
    # scoring mode on the produced sample reproduces the same log-prob
    _, lp2 = d.step(st, e, t, x, prev_sample=xn, eta=1.0)
    np.testing.assert_allclose(lp2, lp, rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        d.step(st, e, t, x, noise=z, prev_sample=xn, eta=1.0)


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("train_cfg", [True, False])
def test_ppo_closed_form_matches_autograd(pred, train_cfg):
    d = DDIMOracle(prediction_type=pred)
    st = d.set_timesteps(d.create_state(), 50)
    g = torch.Generator().manual_seed(1)
    B = 6
    shp = (B, 4, 8, 8)
    eps_c = torch.randn(shp, generator=g, dtype=torch.float64)
    eps_u = torch.randn(shp, generator=g, dtype=torch.float64)
    lat = torch.randn(shp, generator=g, dtype=torch.float64)
    ts = torch.tensor([981, 481, 1, 21, 701, 241])
    # build next_latents near the posterior mean so that ratios straddle the clip range
    with torch.no_grad():
        guided = eps_u + 5.0 * (eps_c - eps_u) if train_cfg else eps_c
        nxt, _ = d.step(st, guided.numpy().astype(np.float32), ts.numpy(), lat.numpy().astype(np.float32),
                        noise=torch.randn(shp, generator=g).numpy(), eta=1.0)
    nxt = torch.from_numpy(nxt).double()
    batch = {"latents": lat, "next_latents": nxt, "ts": ts.numpy(),
             "advantages": torch.tensor([1.5, -0.7, 12.0, -20.0, 0.3, -0.2], dtype=torch.float64)}
    ec = eps_c.clone().requires_grad_(True)
    eu = eps_u.clone().requires_grad_(True)
    with torch.no_grad():
        lp0 = ppo.log_prob_torch(d, st, (eu + 5.0 * (ec - eu)) if train_cfg else ec, batch["ts"], lat, nxt, 1.0, torch.float64)
    # old log-probs: inside, above and below the clip range
    batch["log_probs"] = lp0 + torch.tensor([0.0, 5e-5, -3e-4, 3e-4, 3e-4, -3e-4], dtype=torch.float64)
    loss, info, lp = ppo.loss_and_info_torch(d, st, ec, eu, batch, 5.0, 1.0, 1e-4, train_cfg, torch.float64)
    loss.backward()
    l2, info2, lp2, dc, du = ppo.closed_form_numpy(d, st, eps_c.numpy().astype(np.float32), eps_u.numpy().astype(np.float32),
                                                   lat.numpy().astype(np.float32), nxt.numpy().astype(np.float32), ts.numpy(),
                                                   batch["log_probs"].numpy().astype(np.float32), batch["advantages"].numpy(),
                                                   5.0, 1.0, 1e-4, train_cfg)
    np.testing.assert_allclose(lp2, lp.detach().numpy(), rtol=2e-5, atol=2e-5)
    assert abs(float(l2) - float(loss)) < 1e-4 * max(1, abs(float(loss)))
    assert float(info2["clipfrac"]) == pytest.approx(float(info["clipfrac"]))
    scale = float(ec.grad.abs().max())
    np.testing.assert_allclose(dc, ec.grad.numpy(), rtol=2e-3, atol=2e-4 * scale)
    if train_cfg:
        np.testing.assert_allclose(du, eu.grad.numpy(), rtol=2e-3, atol=2e-4 * scale)
    # clipped samples carry exactly zero gradient
    clipped_rows = (ec.grad.flatten(1).abs().sum(1) == 0).numpy()
    assert clipped_rows.any() and (np.abs(dc).reshape(B, -1).sum(1)[clipped_rows] == 0).all()


def test_bf16_round_matches_torch():
    x = np.random.default_rng(0).standard_normal(10000).astype(np.float32) * 10
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(bf16_round(x), want)


def test_adamw_against_float64_rederivation():
    rng = np.random.default_rng(0)
    p = [rng.standard_normal((37, 5)).astype(np.float32), rng.standard_normal(11).astype(np.float32)]
    opt = AdamWBf16Mu(mu_decay_in_bf16=False)
    st = AccumulatingState(p, opt)
    m64 = [np.zeros_like(x, dtype=np.float64) for x in p]
    v64 = [np.zeros_like(x, dtype=np.float64) for x in p]
    p64 = [x.astype(np.float64) for x in p]
    for t in range(1, 4):
        g1 = [rng.standard_normal(x.shape).astype(np.float32) * 3 for x in p]
        g2 = [rng.standard_normal(x.shape).astype(np.float32) * 3 for x in p]
        st.apply_gradients(g1, do_update=False)
        assert st.n_acc == 1
        st.apply_gradients(g2, do_update=True)
        assert st.n_acc == 0 and all((ga == 0).all() for ga in st.grad_acc) and st.step == t
        g = [(a.astype(np.float64) + b) / 2 for a, b in zip(g1, g2)]
        n = math.sqrt(sum((x ** 2).sum() for x in g))
        assert abs(float(st.last_grad_norm) - n) < 1e-5 * n
        if n >= 1.0:
            g = [x / n for x in g]
        for i in range(len(p)):
            m64[i] = 0.9 * bf16_round(m64[i].astype(np.float32)).astype(np.float64) + 0.1 * g[i]
            v64[i] = 0.999 * v64[i] + 0.001 * g[i] ** 2
            u = (m64[i] / (1 - 0.9 ** t)) / (np.sqrt(v64[i] / (1 - 0.999 ** t)) + 1e-8) + 1e-4 * p64[i]
            p64[i] = p64[i] - 1e-5 * u
            np.testing.assert_allclose(st.params[i], p64[i], rtol=1e-6, atol=1e-7)


def test_accumulate_then_reduce_equals_reduce_then_accumulate():
    """mean-over-ranks and sum-over-steps commute (justifies one all-reduce per optimizer update)."""
    rng = np.random.default_rng(1)
    g = rng.standard_normal((2, 3, 50))          # (rank, micro-step, param)
    a = g.mean(0).sum(0)
    b = g.sum(1).mean(0)
    np.testing.assert_allclose(a, b, rtol=1e-12)
    assert float(global_norm([a.astype(np.float32)])) == pytest.approx(float(np.sqrt((a ** 2).sum())), rel=1e-6)


def test_key_tree_shapes():
    keys = prng.sample_key_tree(0, n_devices=8, n_batches=2)
    assert len(keys) == 2 and keys[0].shape == (8, 2) and keys[0].dtype == np.uint32
    init, zs = prng.device_noise_stream(keys[0][3], (2, 4, 8, 8), 4)
    assert init.shape == (2, 4, 8, 8) and len(zs) == 4 and not np.array_equal(zs[0], zs[1])


def test_bf16x3_arithmetic_model_level_error_budget(monkeypatch):
    """Why the GPU's bf16x3 datapath (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on bf16 MFMA, fp32 accumulate) is admissible and a
    single bf16 pass is not: the same arithmetic emulated on the CPU oracle (every conv / dense of the U-Net), measured against
    float64.  On the full SD-1.5 architecture (random init, 32x32 latents) the emulation gives 2.0e-5 rms (plain fp32: 1.2e-6,
    fp16-rounded weights: 1.0e-3, one bf16 pass: 1.1e-2); the tiny architecture below pins the same ordering in a second."""
    import torch.nn.functional as TF
    from oracle import unet as OU
    cfg = OU.TINY
    p = OU.init_params(OU.unet_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 16, 16, generator=g)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    t = torch.tensor([481, 21], dtype=torch.int32)
    hi = lambda v: v.bfloat16().float()

    def split(v):
        h = hi(v)
        return h, hi(v - h)

    def make(npass):
        def conv(p_, name, xx, stride=1, pad=1):
            w = p_[name + ".kernel"].permute(3, 2, 0, 1)
            (xh, xl), (wh, wl) = split(xx), split(w)
            y = TF.conv2d(xh, wh, None, stride=stride, padding=pad)
            if npass == 3:
                y = TF.conv2d(xl, wh, None, stride=stride, padding=pad) + TF.conv2d(xh, wl, None, stride=stride, padding=pad) + y
            return y + p_[name + ".bias"][None, :, None, None]

        def dense(p_, name, xx):
            (xh, xl), (wh, wl) = split(xx), split(p_[name + ".kernel"])
            y = xh @ wh
            if npass == 3:
                y = xl @ wh + xh @ wl + y
            b = p_.get(name + ".bias")
            return y if b is None else y + b
        return conv, dense

    with torch.no_grad():
        ref = OU.unet_forward({k: v.double() for k, v in p.items()}, cfg, x.double(), t, ctx.double())
        fp32 = OU.unet_forward(p, cfg, x, t, ctx)
        errs = {}
        for npass in (3, 1):
            conv, dense = make(npass)
            monkeypatch.setattr(OU, "_conv2d", conv)
            monkeypatch.setattr(OU, "_dense_f", dense)
            out = OU.unet_forward(p, cfg, x, t, ctx)
            errs[npass] = float((out.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    e32 = float((fp32.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert e32 < 1e-5
    assert errs[3] < 1e-4                       # ~1e-5: an order of magnitude inside the 1e-3 tolerance of north_star
    assert errs[1] > 1e-3 > 10 * errs[3]        # one pass (what XLA's TPU default does) is outside it


def test_f16mx_arithmetic_model_level_error_budget(monkeypatch):
    """The opt-in f16mx datapath emulated on the CPU oracle: a*b ~= a_h*b_h + a_h8*b_l8 + a_l8*b_h8 with h = f16(x), l = x - h, activations'
    8-bit parts e5m2 at a fixed scale (h8 = e5m2(h), l8 = e5m2(l * 2^11) / 2^11), weights' 8-bit parts e4m3 with one power-of-two scale per
    output column (ddpo_amd/csrc/common.h, ddpo_pack_weights_f16mx) — applied to EVERY conv / dense of the tiny U-Net (the GPU routes only the
    long reductions through it, so this bounds it from above).  On SD-1.5 (random init, 32x32 latents) the same emulation gave 7.0e-5 rms against
    float64 (bf16x3 2.0e-5) and the GPU kernel measured 7.2e-5 / 4.2e-5 (all eligible layers / K >= 2560 only, tests/test_gpu_f16mx_model.py);
    the tiny architecture pins the ordering bf16x3 < f16mx << 1e-3 << one bf16 pass."""
    import torch.nn.functional as TF
    from oracle import unet as OU
    cfg = OU.TINY
    p = OU.init_params(OU.unet_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 16, 16, generator=g)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    t = torch.tensor([481, 21], dtype=torch.int32)
    E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2

    def a_parts(v):                                  # activations: fixed scale
        h = v.half().float()
        return h, h.clamp(-57344, 57344).to(E5).float(), ((v - h) * 2048.0).clamp(-57344, 57344).to(E5).float() / 2048.0

    def w_parts(w, dim):                             # weights (reduction along `dim`): per-column power-of-two scale, max|w| / s in [128, 256)
        h = w.half().float()
        amax = w.abs().amax(dim, keepdim=True).clamp_min(2.0 ** -95)
        s = torch.pow(2.0, torch.floor(torch.log2(amax)) - 7)
        q = lambda v: (v / s).clamp(-448, 448).to(E4).float() * s
        return h, q(h), q((w - h) * 2048.0) / 2048.0

    def conv(p_, name, xx, stride=1, pad=1):
        w = p_[name + ".kernel"].permute(3, 2, 0, 1)
        b = p_[name + ".bias"][None, :, None, None]
        O = w.shape[0]
        xh, xh8, xl8 = a_parts(xx)
        wh, wh8, wl8 = (z.reshape(w.shape) for z in w_parts(w.reshape(O, -1), 1))
        c = lambda a, b_: TF.conv2d(a, b_, None, stride=stride, padding=pad)
        return c(xh, wh) + c(xh8, wl8) + c(xl8, wh8) + b

    def dense(p_, name, xx):
        w = p_[name + ".kernel"]
        xh, xh8, xl8 = a_parts(xx)
        wh, wh8, wl8 = w_parts(w, 0)
        y = xh @ wh + xh8 @ wl8 + xl8 @ wh8
        b = p_.get(name + ".bias")
        return y if b is None else y + b

    def bf16x3_conv(p_, name, xx, stride=1, pad=1):
        w = p_[name + ".kernel"].permute(3, 2, 0, 1)
        sp = lambda v: (v.bfloat16().float(), (v - v.bfloat16().float()).bfloat16().float())
        (xh, xl), (wh, wl) = sp(xx), sp(w)
        c = lambda a, b_: TF.conv2d(a, b_, None, stride=stride, padding=pad)
        return c(xl, wh) + c(xh, wl) + c(xh, wh) + p_[name + ".bias"][None, :, None, None]

    def bf16x3_dense(p_, name, xx):
        sp = lambda v: (v.bfloat16().float(), (v - v.bfloat16().float()).bfloat16().float())
        (xh, xl), (wh, wl) = sp(xx), sp(p_[name + ".kernel"])
        y = xl @ wh + xh @ wl + xh @ wh
        b = p_.get(name + ".bias")
        return y if b is None else y + b

    with torch.no_grad():
        ref = OU.unet_forward({k: v.double() for k, v in p.items()}, cfg, x.double(), t, ctx.double())
        errs = {}
        for tag, (c, d) in (("f16mx", (conv, dense)), ("bf16x3", (bf16x3_conv, bf16x3_dense))):
            monkeypatch.setattr(OU, "_conv2d", c)
            monkeypatch.setattr(OU, "_dense_f", d)
            out = OU.unet_forward(p, cfg, x, t, ctx)
            errs[tag] = float((out.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert errs["bf16x3"] < errs["f16mx"] < 2e-4            # measured: ~2.4e-5 and ~8.5e-5
    assert errs["f16mx"] < 8 * errs["bf16x3"]
    assert 5 * errs["f16mx"] < 1e-3                          # well inside the north-star tolerance
