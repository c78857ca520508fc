"""RWR (reward-weighted regression) train step on the HIP engine against the oracle (oracle/diffusion.py, itself pinned to the reference's
ddpo/training/diffusion.py executed in place: tests/test_reference_rwr_goldens.py).  Reference: /root/reference/ddpo/training/diffusion.py:6-102.
  * ddpo_rwr_noisy_latents + the host key tree (split / normal / randint)  == oracle.diffusion.prepare
  * ddpo_rwr_mse_fwd_bwd  == float64 autograd of the pinned loss (batch mean and reward-weighted sum, with and without CFG training)
  * the whole step (U-Net forward / backward between the two kernels): loss, gradient norm and per-parameter gradients vs float64 autograd through the
    oracle U-Net (tiny, fp32 and bf16x3 datapaths; full-size SD-1.5 on bf16x3), and the applied AdamW update vs the optax restatement."""
import math
import os

import numpy as np
import pytest
import torch

from ddpo_amd import lib as L
from oracle import diffusion as OD, prng as OP, unet as OU

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(autouse=True)
def _restore_datapath():
    old = L.DATAPATH
    yield
    L.DATAPATH = old
    L.PACKED.clear()


def _moments(B, hw, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.randn(B, hw, hw, 8, generator=g)
    m[..., 4:] = m[..., 4:] * 3.0 - 4.0                  # log-variances around -4, some outside the clip range [-30, 20] below
    m[0, 0, 0, 4] = -35.0
    m[-1, -1, -1, 7] = 25.0
    return m


@pytest.mark.parametrize("B,hw", [(1, 8), (3, 16), (8, 64)])
def test_prepare_latents_equals_the_oracle(B, hw):
    from ddpo_amd.training.diffusion import DDPMNoiseScheduler, prepare_latents
    sched = DDPMNoiseScheduler()
    acp = OD.ddpm_alphas_cumprod()
    assert np.array_equal(sched.alphas_cumprod, acp)
    m = _moments(B, hw, B + hw)
    key = OP.PRNGKey(11 * B + hw)
    ref = OD.prepare(m.numpy(), key, acp)
    noise, ts, noisy, new_rng = prepare_latents(m.to(DEV), key, sched.create_state(DEV))
    assert np.array_equal(ts.cpu().numpy(), ref["timesteps"])                       # jax.random.randint: integer-exact
    assert np.array_equal(np.asarray(new_rng), ref["new_train_rng"])
    assert _rel(noise, ref["noise"]) < 2e-6                                          # Threefry words bit-equal; erfinv polynomial in fp32
    assert _rel(noisy, ref["noisy_latents"]) < 2e-6


@pytest.mark.parametrize("train_cfg", [True, False])
@pytest.mark.parametrize("weighted", [False, True])
def test_weighted_mse_loss_and_gradient_equal_float64_autograd(train_cfg, weighted):
    B, C, hw, g_scale = 5, 4, 16, 3.0
    g = torch.Generator().manual_seed(7 + int(train_cfg) + 2 * int(weighted))
    ec, eu, noise = (torch.randn(B, C, hw, hw, generator=g) for _ in range(3))
    w = torch.softmax(torch.randn(B, generator=g), 0) if weighted else None
    ecd = ec.double().requires_grad_(True)
    eud = eu.double().requires_grad_(True)
    loss_ref, per_ref = OD.loss_torch(ecd, eud if train_cfg else None, noise.double(), None if w is None else w.double(), g_scale, train_cfg)
    loss_ref.backward()
    d_c, d_u, per, loss = L.rwr_mse_fwd_bwd(ec.to(DEV), eu.to(DEV) if train_cfg else None, noise.to(DEV), None if w is None else w.to(DEV), g_scale, train_cfg)
    assert float(loss[0]) == pytest.approx(float(loss_ref), rel=2e-6)
    assert _rel(per[:, 0], per_ref) < 2e-6
    assert _rel(d_c, ecd.grad) < 2e-6
    if train_cfg:
        assert _rel(d_u, eud.grad) < 2e-6
    else:
        assert d_u is None


def _rwr_forward_backward(unet, batch, key, sched_state, train_cfg, g_scale, weights):
    """The device part of training.diffusion.train_step up to (not including) the optimizer update, so the accumulated gradients can be read."""
    from ddpo_amd.training.diffusion import prepare_latents
    noise, ts, noisy, _ = prepare_latents(batch["vae"].to(DEV), key, sched_state)
    b = noisy.shape[0]
    tape = []
    emb, unc = batch["prompt_embeds"].to(DEV), batch["uncond_embeds"].to(DEV)
    if train_cfg:
        out = unet.forward(torch.cat([noisy, noisy]), torch.cat([ts, ts]), torch.cat([unc, emb]).contiguous(), tape=tape)
        eps_u, eps_c = out[:b].contiguous(), out[b:].contiguous()
    else:
        eps_u, eps_c = None, unet.forward(noisy, ts, emb.contiguous(), tape=tape)
    w = None if weights is None else torch.as_tensor(weights, dtype=torch.float32, device=DEV)
    d_c, d_u, per, loss = L.rwr_mse_fwd_bwd(eps_c, eps_u, noise, w, g_scale, train_cfg)
    unet.backward(tape, torch.cat([d_u, d_c]) if train_cfg else d_c)
    return float(loss[0])


@pytest.mark.parametrize("datapath,train_cfg,weighted", [("fp32", True, True), ("fp32", False, False), ("bf16x3", True, True), ("bf16x3", True, False)])
def test_rwr_step_gradients_match_float64_autograd_tiny(datapath, train_cfg, weighted):
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    from ddpo_amd.training.diffusion import DDPMNoiseScheduler
    L.DATAPATH = datapath
    B, hw = 3, 8
    op = OU.init_params(OU.unet_param_shapes(OU.TINY), seed=4)
    unet = UNet2DCondition(UNetConfig.named("tiny"), DEV)
    unet.params.load_dict(op)
    if datapath != "fp32":
        unet.params.pack_bf16(bwd=True)
    g = torch.Generator().manual_seed(21)
    batch = {"vae": _moments(B, hw, 5), "prompt_embeds": torch.randn(B, 77, 64, generator=g), "uncond_embeds": torch.randn(1, 77, 64, generator=g).expand(B, -1, -1).contiguous()}
    w = torch.softmax(torch.randn(B, generator=g), 0).numpy() if weighted else None
    key = OP.PRNGKey(99)
    acp = OD.ddpm_alphas_cumprod()
    ograds, oloss, _ = OD.train_step_grads({k: v.double() for k, v in op.items()}, OU.TINY, batch["vae"].numpy(), batch["prompt_embeds"], batch["uncond_embeds"], key, acp,
                                           weights=w, train_cfg=train_cfg, guidance_scale=2.5, dtype=torch.float64)
    loss = _rwr_forward_backward(unet, batch, key, DDPMNoiseScheduler().create_state(DEV), train_cfg, 2.5, w)
    assert loss == pytest.approx(oloss, rel=1e-3)
    G = unet.grads
    gn_o = math.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values()))
    gn = math.sqrt(float((G.flat.double() ** 2).sum()))
    assert gn == pytest.approx(gn_o, rel=1e-3)                      # north-star tolerance on grad norms
    worst = max((_rel(G[n], ograds[n]), n) for n in ograds if float(ograds[n].abs().max()) > 1e-6 * gn_o)
    assert worst[0] < 2e-3, worst


def test_rwr_train_step_applies_the_optax_update_tiny():
    """training.diffusion.train_step end to end (every call is an optimizer step, like flax TrainState.apply_gradients): loss, next key, step counter,
    gradient norm and the applied update vs oracle autograd + the optax AdamW(bf16 mu) restatement."""
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    from ddpo_amd.training.diffusion import DDPMNoiseScheduler, train_step
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig
    from oracle.optim import AdamWBf16Mu, AccumulatingState
    L.DATAPATH = "fp32"
    B, hw = 2, 8
    op = OU.init_params(OU.unet_param_shapes(OU.TINY), seed=8)
    unet = UNet2DCondition(UNetConfig.named("tiny"), DEV)
    unet.params.load_dict(op)
    g = torch.Generator().manual_seed(2)
    batch = {"vae": _moments(B, hw, 9), "prompt_embeds": torch.randn(B, 77, 64, generator=g), "uncond_embeds": torch.randn(1, 77, 64, generator=g).expand(B, -1, -1).contiguous()}
    w = np.array([0.7, 0.3], dtype=np.float32)
    key = OP.PRNGKey(5)
    sched = DDPMNoiseScheduler()
    ograds, oloss, prep = OD.train_step_grads(op, OU.TINY, batch["vae"].numpy(), batch["prompt_embeds"], batch["uncond_embeds"], key, sched.alphas_cumprod,
                                              weights=w, train_cfg=True, guidance_scale=1.0, dtype=torch.float32)
    names = list(op.keys())
    ostate = AccumulatingState([op[n].numpy() for n in names], AdamWBf16Mu())
    ostate.apply_gradients([ograds[n].numpy() for n in names], do_update=True)
    state = AccumulatingTrainState(unet, AdamWConfig())
    dbatch = {k: v.to(DEV) for k, v in batch.items()}
    state, loss, new_rng = train_step(state, None, dbatch, key, sched.create_state(DEV), (sched, None, True, 1.0), weights=w)
    assert float(loss) == pytest.approx(oloss, rel=1e-3)
    assert np.array_equal(np.asarray(new_rng), prep["new_train_rng"])
    assert state.step == 1 and state.n_acc == 0 and float(unet.grads.flat.abs().max()) == 0.0
    assert float(state.last_grad_norm) == pytest.approx(float(ostate.last_grad_norm), rel=1e-3)
    gmax = max(float(v.abs().max()) for v in ograds.values())
    checked = 0
    for n, ref in zip(names, ostate.params):
        mask = (ograds[n].abs() > 1e-3 * gmax).numpy()             # Adam normalises ~0 gradients into round-off-sized steps: skip them
        if not mask.any():
            continue
        checked += 1
        upd = np.abs(ref - op[n].numpy())[mask].max()
        err = np.abs(unet.params[n].cpu().numpy() - ref)[mask].max()
        assert err <= 2e-2 * upd + 1e-9, (n, err, upd)
    assert checked > 50


def test_rwr_step_sd15_full_size_shipped_datapath():
    """One RWR step of the real architecture (SD-1.5, 859.5 M parameters, 64x64 latents = 512^2 images) on the shipped datapath against
    float32 autograd through the oracle U-Net: loss and gradient norm within the north-star 1e-3, gradient vector within 2e-3."""
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    from ddpo_amd.training.diffusion import DDPMNoiseScheduler
    L.DATAPATH = os.environ.get("DDPO_PARITY_DATAPATH") or L.SHIPPED_DATAPATH          # f16mx since round 4
    B, hw = 1, 64
    op = OU.init_params(OU.unet_param_shapes(OU.SD15), seed=1)
    unet = UNet2DCondition(UNetConfig.named("sd15"), DEV)
    unet.params.load_dict(op)
    unet.params.pack_bf16(bwd=True)
    g = torch.Generator().manual_seed(31)
    batch = {"vae": _moments(B, hw, 12), "prompt_embeds": torch.randn(B, 77, 768, generator=g), "uncond_embeds": torch.randn(B, 77, 768, generator=g)}
    key = OP.PRNGKey(1234)
    ograds, oloss, _ = OD.train_step_grads(op, OU.SD15, batch["vae"].numpy(), batch["prompt_embeds"], batch["uncond_embeds"], key, OD.ddpm_alphas_cumprod(),
                                           weights=None, train_cfg=False, guidance_scale=1.0, dtype=torch.float32)
    loss = _rwr_forward_backward(unet, batch, key, DDPMNoiseScheduler().create_state(DEV), False, 1.0, None)
    assert loss == pytest.approx(oloss, rel=1e-3)
    G = unet.grads
    gn_o = math.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values()))
    gn = math.sqrt(float((G.flat.double() ** 2).sum()))
    from conftest import parity_record
    num0 = sum(float((G[n].double().cpu() - ograds[n].double()).pow(2).sum()) for n in ograds)
    parity_record(f"\n[rwr sd15 full size] {L.DATAPATH}: loss {loss:.6f} vs {oloss:.6f}; grad norm {gn:.6e} vs {gn_o:.6e}; ||g-g_ref||/||g_ref|| {math.sqrt(num0) / gn_o:.2e}")
    assert gn == pytest.approx(gn_o, rel=1e-3)
    num = sum(float((G[n].double().cpu() - ograds[n].double()).pow(2).sum()) for n in ograds)
    assert math.sqrt(num) / gn_o < 2e-3
