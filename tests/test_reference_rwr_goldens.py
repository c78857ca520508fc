"""RWR (reward-weighted regression) train step against the reference's own code.

tests/golden/reference_rwr.npz holds what /root/reference/ddpo/training/diffusion.py `train_step`, exec'd UNMODIFIED under the numpy jax
shim (tests/golden/make_reference_rwr_goldens.py), computed for four batches (train_cfg on / off, with / without weights): the key
tree, the posterior sample of the stored VAE moments (logvar clip included), the noise and the randint timesteps, the noisy latents the
U-Net is called with, the text embeddings of prompt / empty prompt and the loss.  CPU: the oracle restatement (oracle/diffusion.py)
must reproduce them.  GPU (`-m gpu`): the HIP product path — Threefry normals drawn in the shapes JAX draws them in, the fused
posterior-sample / add-noise kernel, the weighted-MSE forward / backward kernel — through the C ABI."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion as OD

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_rwr.npz"))
CASES = sorted({k.split("/")[0] for k in G.files if k.startswith("case")})
F = np.float32


def toy_unet(lat, t, ctx, scale=F(0.9), bias=F(0.05)):
    c = ctx.astype(F).mean(axis=(1, 2), dtype=F)
    tt = t.astype(F) / F(1000.0)
    return ((F(0.6) * lat / (F(1.0) + F(0.25) * lat * lat) + F(0.3) * tt[:, None, None, None] + F(0.5) * c[:, None, None, None]) * scale + bias).astype(F)


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_the_reference_run(case):
    g = lambda k: G[f"{case}/{k}"]
    prep = OD.prepare(g("moments"), g("rng"), G["alphas_cumprod"])
    assert np.array_equal(prep["new_train_rng"], g("new_rng"))
    assert np.array_equal(prep["timesteps"], g("timesteps"))                       # integer work: exact
    for k in ("latents", "noise", "noisy_latents"):
        np.testing.assert_allclose(prep[k], g(k), rtol=2e-6, atol=2e-6)
    train_cfg, gs = bool(g("cfg")[0]), float(g("cfg")[1])
    w = g("weights")
    eps_c = torch.from_numpy(toy_unet(prep["noisy_latents"], prep["timesteps"], g("cond_embeds")))
    eps_u = torch.from_numpy(toy_unet(prep["noisy_latents"], prep["timesteps"], g("uncond_embeds"))) if train_cfg else None
    loss, _ = OD.loss_torch(eps_c, eps_u, torch.from_numpy(prep["noise"]), torch.from_numpy(w) if w.size else None, gs, train_cfg)
    assert float(loss) == pytest.approx(float(g("loss")), rel=5e-6)


def test_randint_restatement_is_uniform_and_in_range():
    from oracle import prng as OP
    x = OP.randint(OP.PRNGKey(7), (200000,), 0, 1000)
    assert x.dtype == np.int32 and x.min() == 0 and x.max() == 999
    counts = np.bincount(x, minlength=1000)
    assert abs(counts.std() - np.sqrt(200.0)) < 3.0                                  # Poisson spread of a uniform draw
    assert np.array_equal(OP.randint(OP.PRNGKey(7), (5,), 3, 4), np.full(5, 3, dtype=np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_product_path_matches_the_reference_run(case):
    from ddpo_amd import lib as L
    from ddpo_amd.training import diffusion as PD
    from ddpo_amd.utils import prng as PP
    g = lambda k: G[f"{case}/{k}"]
    dev = "cuda"
    sched = PD.DDPMNoiseScheduler()
    assert np.array_equal(sched.alphas_cumprod, G["alphas_cumprod"])
    acp = sched.create_state(dev)
    noise, ts, noisy, new_rng = PD.prepare_latents(torch.from_numpy(g("moments")).to(dev), g("rng"), acp)
    assert np.array_equal(new_rng, g("new_rng")) and np.array_equal(ts.cpu().numpy(), g("timesteps"))
    assert np.array_equal(PP.randint(PP.split(PP.split(g("rng"), 3)[1])[1], (len(g("timesteps")),), 0, 1000), g("timesteps"))
    np.testing.assert_allclose(noise.cpu().numpy(), g("noise"), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(noisy.cpu().numpy(), g("noisy_latents"), rtol=1e-5, atol=1e-5)
    # loss + closed-form gradient of the weighted MSE: the reference's loss value, float64 autograd for the gradient
    train_cfg, gs = bool(g("cfg")[0]), float(g("cfg")[1])
    w = g("weights")
    e_c = toy_unet(g("noisy_latents"), g("timesteps"), g("cond_embeds"))
    e_u = toy_unet(g("noisy_latents"), g("timesteps"), g("uncond_embeds")) if train_cfg else None
    tw = torch.from_numpy(w).to(dev) if w.size else None
    d_c, d_u, per, loss = L.rwr_mse_fwd_bwd(torch.from_numpy(e_c).to(dev), torch.from_numpy(e_u).to(dev) if train_cfg else None,
                                            torch.from_numpy(g("noise")).to(dev), tw, gs, train_cfg)
    assert float(loss) == pytest.approx(float(g("loss")), rel=1e-5)
    tc = torch.from_numpy(e_c).double().requires_grad_(True)
    tu = torch.from_numpy(e_u).double().requires_grad_(True) if train_cfg else None
    ref, ref_per = OD.loss_torch(tc, tu, torch.from_numpy(g("noise")).double(), torch.from_numpy(w).double() if w.size else None, gs, train_cfg)
    ref.backward()
    np.testing.assert_allclose(per[:, 0].cpu().numpy(), ref_per.detach().numpy(), rtol=2e-6)
    np.testing.assert_allclose(d_c.cpu().numpy(), tc.grad.numpy(), rtol=1e-5, atol=1e-9)
    if train_cfg:
        np.testing.assert_allclose(d_u.cpu().numpy(), tu.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("train_cfg,weighted", [(True, True), (False, False)])
def test_rwr_train_step_gradients_match_the_oracle_unet(train_cfg, weighted):
    """Whole step on the tiny U-Net: prepare -> U-Net forward (cond + uncond as one batch) -> weighted MSE -> U-Net backward; parameter
    gradients against torch autograd through the oracle U-Net in float64, loss against its value; then the optimizer update moves the
    parameters (one fused clip + AdamW step, nothing accumulated)."""
    import math
    from ddpo_amd import lib as L
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    from ddpo_amd.training import diffusion as PD
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig
    from oracle import unet as OU, prng as OP
    old = L.DATAPATH
    L.DATAPATH = "bf16x3"
    try:
        op = OU.init_params(OU.unet_param_shapes(OU.TINY), seed=2)
        unet = UNet2DCondition(UNetConfig.named("tiny"), "cuda")
        unet.params.load_dict(op)
        unet.params.pack_bf16()
        g = torch.Generator().manual_seed(4)
        B = 3
        moments = torch.cat([torch.randn(B, 8, 8, 4, generator=g) * 0.8, torch.randn(B, 8, 8, 4, generator=g) * 0.5 - 1.0], dim=-1)
        emb, unc = torch.randn(B, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g).expand(B, -1, -1).contiguous()
        weights = (torch.rand(B, generator=g) / B) if weighted else None
        rng = OP.PRNGKey(11)
        sched = PD.DDPMNoiseScheduler()
        ograds, oloss, prep = OD.train_step_grads(op, OU.TINY, moments.numpy(), emb, unc, rng, sched.alphas_cumprod,
                                                  None if weights is None else weights.numpy(), train_cfg, 3.0, dtype=torch.float64)
        state = AccumulatingTrainState(unet, AdamWConfig(learning_rate=1e-4))
        captured = {}

        def fake_apply(do_update):                  # read the gradients before the fused update zeroes them
            captured["g"] = {n: unet.grads[n].clone() for n in ograds}
            return AccumulatingTrainState.apply_gradients(state, do_update=do_update)
        state.apply_gradients = fake_apply
        before = unet.params.flat.clone()
        batch = {"vae": moments.to("cuda"), "prompt_embeds": emb.to("cuda"), "uncond_embeds": unc.to("cuda")}
        state, loss, new_rng = PD.train_step(state, None, batch, rng, sched.create_state("cuda"), (sched, None, train_cfg, 3.0), weights=weights)
        assert np.array_equal(new_rng, prep["new_train_rng"])
        assert float(loss) == pytest.approx(oloss, rel=1e-3)
        gn_o = math.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values()))
        num = sum(float(((captured["g"][n].cpu().double() - ograds[n].double()) ** 2).sum()) for n in ograds)
        print(f"\n[rwr train step] tiny cfg={train_cfg} weighted={weighted}: loss rel {abs(float(loss) - oloss) / abs(oloss):.2e}  ||g-g_ref||/||g_ref|| {math.sqrt(num) / gn_o:.2e}")
        assert math.sqrt(num) / gn_o < 2e-3
        assert state.step == 1 and state.n_acc == 0 and not torch.equal(unet.params.flat, before)
        assert float(unet.grads.flat.abs().max()) == 0.0
    finally:
        L.DATAPATH = old
        L.PACKED.clear()
