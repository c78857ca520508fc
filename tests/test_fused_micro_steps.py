"""Fused PPO micro-steps (`train_steps_fused`, `ddpo_ddim_logprob_ppo_fwd_bwd_grouped`).

The reference runs the micro-steps of one accumulation window one at a time and sums their gradients
(/root/reference/ddpo/training/policy_gradient.py:32-48; loop /root/reference/pipeline/policy_gradient.py:407-441).
The CPU test pins the identity the fused path relies on with the ORACLE (sum of per-micro-batch gradients == gradient of
the grouped loss over the concatenated rows); the GPU tests hold the grouped kernel and `train_steps_fused` to k separate
`train_step` calls.

The entrypoint fuses DDPO_TRAIN_FUSE (default 10) timesteps of a mini-batch per launch.
"""
import numpy as np
import pytest
import torch

from oracle import ppo as OPPO
from oracle.ddim import DDIMOracle



def _case(k, b, seed=0, pred="epsilon"):
    rng = np.random.default_rng(seed)
    dd = DDIMOracle(prediction_type=pred)
    ost = dd.set_timesteps(dd.create_state(), 50)
    B, shape = k * b, (k * b, 4, 8, 8)
    ec, eu, x, z = (rng.standard_normal(shape, dtype=np.float32) for _ in range(4))
    ts = rng.choice(np.asarray(ost.timesteps), size=B).astype(np.int32)
    guided = (eu + np.float32(5.0) * (ec - eu)).astype(np.float32)
    xn, lp0 = dd.step(ost, guided, ts, x, noise=z, eta=1.0)
    # old log-probs well inside (3e-5) or well outside (3e-4) the 1e-4 clip range: no fp32-vs-fp64 flips of the clip decision
    old = (lp0 + rng.choice(np.asarray([3e-5, -3e-5, 3e-4, -3e-4], dtype=np.float32), size=B)).astype(np.float32)
    adv = (rng.standard_normal(B) * 2).astype(np.float32)
    adv[0] = 14.0                                                   # exercises ADV_CLIP_MAX
    return dd, ost, ec, eu, x, xn, ts, old, adv


def _loop_reference(dd, ost, ec, eu, x, xn, ts, old, adv, k, b, train_cfg=True):
    """k separate micro-batches through the oracle's closed form (what k train_step calls compute)."""
    dcs, dus, lps, infos = [], [], [], []
    for j in range(k):
        sl = slice(j * b, (j + 1) * b)
        loss, info, lp, dc, du = OPPO.closed_form_numpy(dd, ost, ec[sl], eu[sl], x[sl], xn[sl], ts[sl], old[sl], adv[sl],
                                                        5.0, 1.0, 1e-4, train_cfg)
        dcs.append(dc); dus.append(du); lps.append(lp)
        infos.append([float(info["approx_kl"]), float(info["clipfrac"]), float(info["loss"])])
    return np.concatenate(dcs), np.concatenate(dus), np.concatenate(lps), np.asarray(infos, dtype=np.float32)


@pytest.mark.parametrize("k,b", [(1, 2), (4, 2), (5, 1), (3, 4)])
def test_oracle_grouped_loss_is_sum_of_micro_batch_losses(k, b):
    """autograd of sum_j mean_{rows of micro-batch j}(ppo loss) over the concatenated rows == the per-micro-batch closed
    forms stacked: the identity that lets k micro-steps share one forward/backward (float64 autograd as ground truth)."""
    dd, ost, ec, eu, x, xn, ts, old, adv = _case(k, b, seed=k * 10 + b)
    odc, odu, olp, oinfo = _loop_reference(dd, ost, ec, eu, x, xn, ts, old, adv, k, b)
    tec = torch.from_numpy(ec).double().requires_grad_(True)
    teu = torch.from_numpy(eu).double().requires_grad_(True)
    total = 0.0
    for j in range(k):
        sl = slice(j * b, (j + 1) * b)
        batch = {"ts": ts[sl], "latents": torch.from_numpy(x[sl]), "next_latents": torch.from_numpy(xn[sl]),
                 "advantages": torch.from_numpy(adv[sl]), "log_probs": torch.from_numpy(old[sl])}
        loss, info, lp = OPPO.loss_and_info_torch(dd, ost, tec[sl], teu[sl], batch, 5.0, 1.0, 1e-4, True, dtype=torch.float64)
        total = total + loss
        assert float(loss) == pytest.approx(float(oinfo[j, 2]), rel=1e-4, abs=1e-6)
    total.backward()
    scale = np.abs(odc).max()
    np.testing.assert_allclose(tec.grad.numpy(), odc, rtol=2e-3, atol=2e-4 * scale)
    np.testing.assert_allclose(teu.grad.numpy(), odu, rtol=2e-3, atol=2e-4 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("k,b,train_cfg", [(4, 2, True), (5, 1, True), (3, 4, False), (1, 6, True)])
def test_grouped_ppo_kernel_matches_separate_micro_batches(pred, k, b, train_cfg):
    from ddpo_amd import lib as L
    from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
    dd, ost, ec, eu, x, xn, ts, old, adv = _case(k, b, seed=3, pred=pred)
    if not train_cfg:                                               # next_latents must come from the unguided prediction then
        xn, lp0 = dd.step(ost, ec, ts, x, noise=np.random.default_rng(9).standard_normal(x.shape, dtype=np.float32), eta=1.0)
        old = (lp0 + np.tile(np.asarray([3e-5, -3e-4, 3e-4, -3e-5], dtype=np.float32), x.shape[0])[:x.shape[0]]).astype(np.float32)
    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1,
                      prediction_type=pred)
    st = s.set_timesteps(s.create_state(device="cuda"), 50)
    consts = s.kernel_consts(st, 1.0)
    t = lambda a: torch.from_numpy(a).to("cuda")
    args = (t(ec), t(eu) if train_cfg else None, t(x), t(xn), t(ts), t(old), t(adv), 5.0, 1e-4, train_cfg, consts)
    d_c, d_u, per, info = L.ddim_logprob_ppo_fwd_bwd(*args, group=b)
    assert info.shape == (k, 3)
    # (1) bit-identical to k separate launches of the ungrouped entry point
    for j in range(k):
        sl = slice(j * b, (j + 1) * b)
        a = [v[sl].contiguous() if torch.is_tensor(v) else v for v in args]
        dc_j, du_j, per_j, info_j = L.ddim_logprob_ppo_fwd_bwd(*a)
        assert torch.equal(dc_j, d_c[sl]) and torch.equal(per_j, per[sl]) and torch.equal(info_j, info[j])
        if train_cfg:
            assert torch.equal(du_j, d_u[sl])
    # (2) and equal to the oracle loop
    odc, odu, olp, oinfo = _loop_reference(dd, ost, ec, eu, x, xn, ts, old, adv, k, b, train_cfg)
    np.testing.assert_allclose(per[:, 0].cpu().numpy(), olp, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(info[:, 2].cpu().numpy(), oinfo[:, 2], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(info[:, 1].cpu().numpy(), oinfo[:, 1], atol=1e-6)
    scale = np.abs(odc).max()
    np.testing.assert_allclose(d_c.cpu().numpy(), odc, rtol=2e-3, atol=2e-4 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("datapath", ["fp32", "bf16x3"])
@pytest.mark.parametrize("jit", [False, True])
def test_train_steps_fused_matches_separate_train_steps(datapath, jit):
    """k = 3 micro-steps of 2 samples (train_cfg), accumulate-only and then with the closing optimizer update:
    same n_acc / step bookkeeping, same info rows, gradients and updated parameters equal up to fp32 summation order."""
    from ddpo_amd import lib as L
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig, train_step, train_steps_fused
    from oracle import unet as OU
    L.DATAPATH = datapath
    k, b, hw = 3, 2, 8
    op = OU.init_params(OU.unet_param_shapes(OU.TINY), seed=4)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
    st = sched.set_timesteps(sched.create_state(device="cuda"), 50)
    g = torch.Generator().manual_seed(11)
    emb = torch.randn(b, 77, 64, generator=g).cuda()
    unc = torch.randn(1, 77, 64, generator=g).expand(b, -1, -1).contiguous().cuda()
    batches = []
    for j in range(k):
        lat = torch.randn(b, 4, hw, hw, generator=g)
        batches.append({"latents": lat.cuda(), "next_latents": (0.95 * lat + 0.1 * torch.randn(lat.shape, generator=g)).cuda(),
                        "ts": torch.tensor([[481, 21], [961, 241], [1, 701]][j], dtype=torch.int32).cuda(),
                        "log_probs": torch.tensor([-1.2, -0.9]).cuda() - 0.01 * j, "advantages": torch.tensor([0.7, -1.1]).cuda(),
                        "prompt_embeds": emb, "uncond_embeds": unc})

    def fresh():
        unet = UNet2DCondition(UNetConfig.named("tiny"), "cuda")
        unet.params.load_dict(op)
        if datapath != "fp32":
            unet.params.pack_bf16()
        return unet, AccumulatingTrainState(unet, AdamWConfig(learning_rate=1e-3))

    for closing_update in (False, True):
        unet_a, sa = fresh()
        infos_a = []
        for j in range(k):
            sa, info = train_step(sa, batches[j], st, sched, True, 5.0, 1.0, 10.0, do_opt_update=(closing_update and j == k - 1), jit=jit)
            infos_a.append(info)
        ga = unet_a.grads.flat.clone()
        unet_b, sb = fresh()
        sb, infos_b = train_steps_fused(sb, batches, st, sched, True, 5.0, 1.0, 10.0, do_opt_update=closing_update, jit=jit)
        gb = unet_b.grads.flat
        assert (sa.n_acc, sa.step, sa.opt_state["count"]) == (sb.n_acc, sb.step, sb.opt_state["count"])
        assert len(infos_b) == k
        for ia, ib in zip(infos_a, infos_b):
            for key in ("approx_kl", "clipfrac", "loss"):
                assert float(ib[key]) == pytest.approx(float(ia[key]), rel=1e-5, abs=1e-7), key
            # forwards are batch-composition independent (fixed reduction orders): the log-probs are the same bits
            assert torch.equal(ia["log_prob"], ib["log_prob"])
        if closing_update:
            assert float(gb.abs().max()) == 0.0 and float(ga.abs().max()) == 0.0
            assert float(sb.last_grad_norm) == pytest.approx(float(sa.last_grad_norm), rel=1e-4)
            # Adam's first step is ~lr * sign(g) whatever |g| is, so entries whose gradient is round-off noise move by
            # +-lr at random: compare the applied update only where the accumulated gradient is well resolved
            upd = max(float((unet_a.params[n].cpu() - op[n]).abs().max()) for n in op)       # size of the applied update
            diff = (unet_a.params.flat - unet_b.params.flat).abs()
            assert int(resolved.sum()) > 1000
            assert float(diff[resolved].max()) <= 2e-2 * upd + 1e-9
        else:
            gn = float(ga.double().norm())
            resolved = ga.abs() > 1e-3 * ga.abs().max()           # used by the closing-update pass below (same batches)
            assert float((ga - gb).double().norm()) <= 2e-5 * gn, (float((ga - gb).double().norm()), gn)
