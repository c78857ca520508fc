"""DDIM scheduler / sampler-loop parity against outputs of the REFERENCE'S OWN CODE.

tests/golden/reference_ddim_sampler.npz was produced by tests/golden/make_reference_ddim_goldens.py, which exec's the
reference's scheduling_ddim_flax.py unmodified (numpy stand-ins for jax.numpy / flax / diffusers base classes) and runs
the lifted body of FlaxStableDiffusionPipeline._generate with a closed-form toy U-Net.  Here:
  * CPU: the oracle restatement (oracle/ddim.py, oracle/sampler.py) must reproduce those outputs — this PINS the oracle;
  * GPU: the product (HIP Threefry + DDIM-step / log-prob kernels through the C ABI, the Python sampler mirror) must too.
Integers (timesteps, ts) are compared exactly; floats with 2e-6 relative tolerance on CPU (numpy `x ** 0.5` = powf vs
sqrt, <= 1 ulp), 1e-5 on the GPU path (erfinv / log / fma contraction differences), log-probs 1e-4 absolute."""
import os

import numpy as np
import pytest
import torch

from oracle import prng as OP
from oracle.ddim import DDIMOracle
from oracle.sampler import sample as oracle_sample

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_ddim_sampler.npz"))
F = np.float32
CASES = sorted({k.rsplit("/", 1)[0] for k in G.files if k.endswith("/score_logp")})
GEN = sorted({k.rsplit("/", 1)[0] for k in G.files if k.startswith("generate/") and k.endswith("/final")})


def _parse(tag):
    ptype, T, s, eta = tag.split("/")
    return ptype, int(T[1:]), int(s[1:]), float(eta[3:])


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def toy_unet_numpy(lat, t, ctx):
    lat = lat.astype(F)
    c = ctx.astype(F).mean(axis=(1, 2), dtype=F)
    tt = t.astype(F) / F(1000.0)
    return (F(0.6) * lat / (F(1.0) + F(0.25) * lat * lat) + F(0.3) * tt[:, None, None, None] + F(0.5) * c[:, None, None, None]).astype(F)


# ------------------------------------------------------------------------------------------------ CPU: oracle is pinned
@pytest.mark.parametrize("ptype", ["epsilon", "v_prediction"])
def test_oracle_schedule_matches_reference_run(ptype):
    dd = DDIMOracle(prediction_type=ptype)
    st = dd.create_state()
    np.testing.assert_allclose(st.alphas_cumprod, G[f"{ptype}/alphas_cumprod"], rtol=2e-6)
    assert float(st.final_alpha_cumprod) == pytest.approx(float(G[f"{ptype}/final_alpha_cumprod"]), rel=2e-6)
    for T in (4, 50):
        assert np.array_equal(dd.set_timesteps(st, T).timesteps, G[f"{ptype}/T{T}/timesteps"])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_step_matches_reference_run(tag):
    ptype, T, si, eta = _parse(tag)
    dd = DDIMOracle(prediction_type=ptype)
    st = dd.set_timesteps(dd.create_state(), T)
    eps, x = G[tag + "/eps"], G[tag + "/x"]
    z = OP.normal(G[tag + "/key"], eps.shape)
    prev, lp = dd.step(st, eps, int(st.timesteps[si]), x, noise=z, eta=eta)              # sampling mode
    assert _rel(prev, G[tag + "/prev"]) < 2e-6
    np.testing.assert_allclose(lp, G[tag + "/logp"], rtol=2e-5, atol=2e-5)
    _, lp2 = dd.step(st, eps, G[tag + "/score_ts"], x, prev_sample=G[tag + "/score_prev"], eta=eta)   # scoring mode, per-sample t
    np.testing.assert_allclose(lp2, G[tag + "/score_logp"], rtol=2e-4, atol=2e-4)


def _gen_args(tag):
    name = tag.split("/", 1)[1]
    ptype = "v_prediction" if name.startswith("v_prediction") else "epsilon"
    parts = name[len(ptype) + 1:].split("_")
    return ptype, int(parts[0][1:]), float(parts[1][1:]), float(parts[2][3:]), int(parts[3][4:])


@pytest.mark.parametrize("tag", GEN)
def test_oracle_sampler_loop_matches_reference_generate(tag):
    ptype, T, g, eta, seed = _gen_args(tag)
    dd = DDIMOracle(prediction_type=ptype)
    cfg = type("Cfg", (), {"in_channels": 4})()
    emb, neg = torch.from_numpy(G[tag + "/emb"]), torch.from_numpy(G[tag + "/neg"])
    final, lat, nxt, lps, ts = oracle_sample(None, cfg, dd, dd.create_state(), emb, neg, OP.PRNGKey(seed), T, 64, 64, g, eta,
                                             unet_fn=toy_unet_numpy)
    assert np.array_equal(ts, G[tag + "/ts"])
    assert _rel(lat[:, 0], G[tag + "/latents"][:, 0]) < 1e-6                   # initial noise: key tree + Threefry + erfinv
    assert _rel(final, G[tag + "/final"]) < 5e-6
    assert _rel(lat, G[tag + "/latents"]) < 5e-6 and _rel(nxt, G[tag + "/next_latents"]) < 5e-6
    np.testing.assert_allclose(lps, G[tag + "/log_probs"], rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------------------------------ GPU: the product
def _product_scheduler(ptype):
    from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
    return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False,
                         steps_offset=1, prediction_type=ptype)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_product_scheduler_step_matches_reference_run(tag):
    ptype, T, si, eta = _parse(tag)
    sch = _product_scheduler(ptype)
    st = sch.set_timesteps(sch.create_state(device="cuda"), T)
    assert np.array_equal(np.asarray(st.timesteps), G[f"{ptype}/T{T}/timesteps"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    eps, x = dev(G[tag + "/eps"]), dev(G[tag + "/x"])
    prev, _, lp = sch.step(st, eps, int(np.asarray(st.timesteps)[si]), x, G[tag + "/key"], None, eta)
    assert _rel(prev.cpu().numpy(), G[tag + "/prev"]) < 1e-5
    np.testing.assert_allclose(lp.cpu().numpy(), G[tag + "/logp"], rtol=1e-4, atol=1e-4)
    _, _, lp2 = sch.step(st, eps, dev(G[tag + "/score_ts"]), x, None, dev(G[tag + "/score_prev"]), eta)
    np.testing.assert_allclose(lp2.cpu().numpy(), G[tag + "/score_logp"], rtol=1e-3, atol=1e-3)
    with pytest.raises(ValueError):
        sch.step(st, eps, 1, x, G[tag + "/key"], x, eta)


class _ToyUNet:
    """Closed-form U-Net stand-in on the GPU (same formula as the fixture's), with the attributes the pipeline reads."""

    def __init__(self):
        self.device = torch.device("cuda")
        self.cfg = type("Cfg", (), {"in_channels": 4})()

    def __call__(self, lat, t, ctx):
        c = ctx.mean(dim=(1, 2))
        tt = t.to(torch.float32) / 1000.0
        return 0.6 * lat / (1.0 + 0.25 * lat * lat) + 0.3 * tt[:, None, None, None] + 0.5 * c[:, None, None, None]

    forward_graphed = __call__


@pytest.mark.gpu
@pytest.mark.parametrize("tag", GEN)
def test_product_pipeline_matches_reference_generate(tag):
    from ddpo_amd.diffusers_patch.pipeline_stable_diffusion import StableDiffusionPipeline
    ptype, T, g, eta, seed = _gen_args(tag)
    sch = _product_scheduler(ptype)
    pipe = StableDiffusionPipeline(_ToyUNet(), None, sch)
    state = sch.create_state(device="cuda")
    emb, neg = torch.from_numpy(G[tag + "/emb"]).cuda(), torch.from_numpy(G[tag + "/neg"]).cuda()
    final, lat, nxt, lps, ts = pipe(emb, neg, {"unet": None, "scheduler": state}, OP.PRNGKey(seed), T, height=64, width=64,
                                    guidance_scale=g, eta=eta)
    assert np.array_equal(ts.cpu().numpy(), G[tag + "/ts"])
    assert _rel(lat[:, 0].cpu().numpy(), G[tag + "/latents"][:, 0]) < 2e-6
    assert _rel(final.cpu().numpy(), G[tag + "/final"]) < 2e-5
    assert _rel(lat.cpu().numpy(), G[tag + "/latents"]) < 2e-5 and _rel(nxt.cpu().numpy(), G[tag + "/next_latents"]) < 2e-5
    np.testing.assert_allclose(lps.cpu().numpy(), G[tag + "/log_probs"], rtol=2e-4, atol=2e-4)


# ------------------------------------------------------------------------------------------------ train_step / accumulation
TRAIN = sorted({k.rsplit("/", 1)[0] for k in G.files if k.startswith("train/") and k.endswith("/loss")})


def _train_args(tag):
    _, ptype, cfg = tag.split("/")
    return ptype, cfg == "cfg1"


@pytest.mark.parametrize("tag", TRAIN)
def test_oracle_ppo_loss_matches_reference_train_step(tag):
    """The reference's train_step (ddpo/training/policy_gradient.py, exec'd unmodified by the fixture generator; jax.grad
    replaced by a value-only stand-in) pins the loss closure: CFG combine, scoring-mode log-prob with per-sample timesteps,
    advantage clip at +-10, ratio, PPO-clip loss, approx_kl, clipfrac."""
    from oracle import ppo
    ptype, train_cfg = _train_args(tag)
    dd = DDIMOracle(prediction_type=ptype)
    st = dd.set_timesteps(dd.create_state(), 50)
    t = lambda k: torch.from_numpy(G[f"{tag}/{k}"])
    batch = {k: t(k) for k in ("latents", "next_latents", "ts", "log_probs", "advantages")}
    loss, info, _ = ppo.loss_and_info_torch(dd, st, t("eps_cond"), t("eps_uncond") if train_cfg else None, batch, 5.0, 1.0, 1e-4, train_cfg)
    assert float(loss) == pytest.approx(float(G[tag + "/loss"]), rel=2e-5)
    assert float(info["clipfrac"]) == pytest.approx(float(G[tag + "/clipfrac"]), abs=1e-7)
    assert float(info["approx_kl"]) == pytest.approx(float(G[tag + "/approx_kl"]), rel=0.05)     # 0.5*mean((1e-4-scale differences of O(1) fp32 numbers)^2)
    out = ppo.closed_form_numpy(dd, st, G[tag + "/eps_cond"], G[tag + "/eps_uncond"], G[tag + "/latents"], G[tag + "/next_latents"],
                                G[tag + "/ts"], G[tag + "/log_probs"], G[tag + "/advantages"], 5.0, 1.0, 1e-4, train_cfg)
    assert float(out[0]) == pytest.approx(float(G[tag + "/loss"]), rel=2e-5)


def test_oracle_accumulation_matches_reference_class():
    """AccumulatingTrainState of the reference (its real class over a minimal TrainState stand-in): n_acc / step / grad_acc
    traces and the averaged gradient the optimizer receives."""
    from oracle.optim import AccumulatingState

    class _SGD:
        def __init__(self):
            self.received = []

        def init(self, params):
            return None

        def update(self, params, grads, state):
            self.received.append(grads[0].copy())
            return [params[0] - grads[0]], state, None
    opt = _SGD()
    acc = AccumulatingState([np.zeros(5, dtype=F)], opt)
    for i, g in enumerate(G["accum/grads"]):
        acc.apply_gradients([g], bool(G["accum/do_update"][i]))
        assert acc.n_acc == int(G["accum/n_acc"][i]) and acc.step == int(G["accum/step"][i])
        np.testing.assert_allclose(acc.grad_acc[0], G["accum/grad_acc"][i], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(acc.params[0], G["accum/params"][i], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.stack(opt.received), G["accum/received"], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TRAIN)
def test_product_ppo_kernel_matches_reference_train_step(tag):
    """ddpo_ddim_logprob_ppo_fwd_bwd (C ABI) on the inputs of the reference-run train_step: loss / approx_kl / clipfrac."""
    from ddpo_amd import lib as L
    ptype, train_cfg = _train_args(tag)
    sch = _product_scheduler(ptype)
    st = sch.set_timesteps(sch.create_state(device="cuda"), 50)
    d = lambda k: torch.from_numpy(np.ascontiguousarray(G[f"{tag}/{k}"])).cuda()
    consts = sch.kernel_consts(st, 1.0)
    _, _, per_sample, info = L.ddim_logprob_ppo_fwd_bwd(d("eps_cond"), d("eps_uncond") if train_cfg else None, d("latents"), d("next_latents"),
                                                        d("ts"), d("log_probs"), d("advantages"), 5.0, 1e-4, train_cfg, consts)
    info = info.cpu().numpy()
    assert float(info[2]) == pytest.approx(float(G[tag + "/loss"]), rel=1e-4)
    assert float(info[1]) == pytest.approx(float(G[tag + "/clipfrac"]), abs=1e-6)
    assert float(info[0]) == pytest.approx(float(G[tag + "/approx_kl"]), rel=0.1)
