"""The opt-in `f16mx` datapath at model level (lib.DATAPATHS): every plane-eligible forward contraction with a long reduction (K >= lib.MX_MIN_K)
on the f16 + MX-fp8 cross-term kernel, everything else as under bf16x3.
  * producers: GroupNorm / LayerNorm / GEMM output stages emit exactly the planes ddpo_split_planes_f16mx makes of their fp32 result;
  * routing: an eligible layer runs the f16mx kernel whether its input arrives as planes or as fp32 (auto-split) — bit-identical outputs, and
    therefore a training forward (fp32 producers, tape) bit-identical to the sampler's forward (plane producers) of the same rows;
  * accuracy: U-Net forward, sampler, PPO train step and RWR-free gradients inside the north-star 1e-3 of the float64 oracle (the datapath's own
    budget is ~1e-4 on a forward: 3-4x bf16x3)."""
import math

import numpy as np
import pytest
import torch

from ddpo_amd import lib as L
from oracle import unet as OU

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(autouse=True)
def _f16mx(monkeypatch):
    old = L.DATAPATH
    L.DATAPATH = "f16mx"
    monkeypatch.setattr(L, "MX_MIN_K", 256)         # the tiny test architecture has no reduction of 2560: make its 3x3 convolutions (K >= 288) f16mx layers
    yield
    L.DATAPATH = old
    L.PACKED.clear()


@pytest.mark.parametrize("silu", [True, False])
def test_norm_producers_emit_the_split_of_their_fp32_result(silu):
    g = torch.Generator(device=DEV).manual_seed(1)
    B, HW, C, G = 3, 100, 96, 32
    x = torch.randn(B * HW, C, device=DEV, generator=g) * 3 + 0.5
    gamma, beta = torch.randn(C, device=DEV, generator=g), torch.randn(C, device=DEV, generator=g)
    pl = L.groupnorm(x, B, HW, gamma, beta, G, 1e-5, silu, planes=2)              # 2 = the planes_pay value of an f16mx consumer
    y = L.groupnorm(x, B, HW, gamma, beta, G, 1e-5, silu)
    ref = L.split_planes(y, fmt=1)
    assert pl.fmt == 1 and ref.fmt == 1 and torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo)
    assert float((pl.float() - y).abs().max() / y.abs().max()) < 2.0 ** -13
    pl0 = L.groupnorm(x, B, HW, gamma, beta, G, 1e-5, silu, planes=1)             # a bf16x3 consumer still gets bf16 hi / lo
    ref0 = L.split_planes(y)
    assert pl0.fmt == 0 and torch.equal(pl0.hi, ref0.hi) and torch.equal(pl0.lo, ref0.lo)
    xl = torch.randn(130, 320, device=DEV, generator=g)
    gl, bl = torch.randn(320, device=DEV, generator=g), torch.randn(320, device=DEV, generator=g)
    pl = L.layernorm(xl, gl, bl, planes=2)
    ref = L.split_planes(L.layernorm(xl, gl, bl), fmt=1)
    assert pl.fmt == 1 and torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo)


@pytest.mark.parametrize("B,H,Cin,Cout,ks", [(2, 16, 320, 96, 3), (2, 32, 320, 320, 3), (1, 8, 1280, 1280, 3), (3, 12, 96, 160, 1), (4, 77, 768, 320, 0), (2, 640, 320, 1280, 0),
                                             (2, 512, 2560, 640, 0), (2, 16, 64, 96, 3)])
def test_eligible_layers_run_f16mx_from_planes_and_from_fp32_alike(B, H, Cin, Cout, ks):
    g = torch.Generator(device=DEV).manual_seed(3)
    conv = ks > 0
    rows = B * H * H if conv else B * H
    K = ks * ks * Cin if conv else Cin
    x = torch.randn(rows, Cin, device=DEV, generator=g)
    w = torch.randn((ks, ks, Cin, Cout) if conv else (Cin, Cout), device=DEV, generator=g) / K ** 0.5
    b = torch.randn(Cout, device=DEV, generator=g)
    L.pack_weights(w, bwd=False)
    run = (lambda s: L.conv2d(s, w, b, B, H, H, Cin, Cout, ks)[0]) if conv else (lambda s: L.linear(s, w, b))
    L.MX_MIN_K = 2560                       # the shipped threshold (the fixture lowers it for the tiny architecture)
    L.PACKED.clear()
    L.pack_weights(w, bwd=False)
    if K < L.MX_MIN_K:                      # short reduction: NOT an f16mx layer — exactly the bf16x3 arithmetic, whatever the feed
        assert not L.mx_layer(w) and L.planes_pay(w, Cin, rows) in (0, 1)
        with L.datapath("bf16x3"):
            y3 = run(x)
        assert torch.equal(run(x), y3) and torch.equal(run(L.split_planes(x)), y3)
        with pytest.raises(L.DdpoHipError):
            run(L.split_planes(x, fmt=1))                                      # f16mx planes handed to a bf16x3 layer: refused, not silently converted
        return
    assert L.mx_layer(w) and L.planes_ok(w, Cin, rows) and L.planes_pay(w, Cin, rows) == 2 and L.planes_pay(w, Cin, 8 * rows) == 2      # layer property
    y_fp32_in = run(x)                      # auto-split on the way in
    y_planes = run(L.split_planes(x, fmt=1))       # planes from a producer
    assert torch.equal(y_fp32_in, y_planes)
    with pytest.raises(L.DdpoHipError):
        run(L.split_planes(x))              # bf16 planes handed to an f16mx layer
    ref = (torch.nn.functional.conv2d(x.view(B, H, H, Cin).permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), None, padding=ks // 2)
           .permute(0, 2, 3, 1).reshape(rows, Cout) if conv else x.double() @ w.double()) + b.double()
    err = float((y_planes.double() - ref).pow(2).mean().sqrt() / (ref - b.double()).pow(2).mean().sqrt())
    assert err < 3e-5
    with L.datapath("bf16x3"):
        y3 = run(x)
    assert not torch.equal(y3, y_planes) and _rel(y3, y_planes) < 3e-4            # really another arithmetic, and close
    # plane-emitting output stage in the datapath's format
    if Cout % 32 == 0 and L.planes_out_ok(w, Cin, rows, Cout):
        kw = dict(M=rows if not conv else B * H * H, N=Cout, K=K, bias=b, planes_out="both", planes_fmt=2)
        if conv:
            kw["conv"] = dict(ksize=ks, stride=1, pad=ks // 2, upsample=0, B=B, H=H, W=H, Cin=Cin, OH=H, OW=H)
        out, opl = L.gemm_conv(x, w, **kw)
        sp = L.split_planes(out, fmt=1)
        assert opl.fmt == 1 and torch.equal(out, y_planes) and torch.equal(opl.hi, sp.hi) and torch.equal(opl.lo, sp.lo)


def _tiny(seed=0):
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    op = OU.init_params(OU.unet_param_shapes(OU.TINY), seed=seed)
    unet = UNet2DCondition(UNetConfig.named("tiny"), DEV)
    unet.params.load_dict(op)
    unet.params.pack_bf16()
    return op, unet


def test_unet_forward_train_forward_and_train_step_tiny():
    from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig, train_step
    from oracle.ddim import DDIMOracle
    from oracle.sampler import train_step_grads
    op, unet = _tiny()
    g = torch.Generator().manual_seed(5)
    b, hw = 4, 8
    lat = torch.randn(b, 4, hw, hw, generator=g)
    ts = torch.tensor([481, 21, 981, 241], dtype=torch.int32)
    emb = torch.randn(b, 77, 64, generator=g)
    unc = torch.randn(1, 77, 64, generator=g).expand(b, -1, -1).contiguous()
    ref = OU.unet_forward({k: v.double() for k, v in op.items()}, OU.TINY, lat.double(), ts, emb.double())
    out = unet(lat.to(DEV), ts.to(DEV), emb.to(DEV))
    e_mx = _rel(out, ref)
    with L.datapath("bf16x3"):
        e_3 = _rel(unet(lat.to(DEV), ts.to(DEV), emb.to(DEV)), ref)
    print(f"\n[f16mx tiny U-Net forward] max rel err {e_mx:.2e} (bf16x3 {e_3:.2e})")
    assert e_mx < 3e-4 and e_3 < e_mx < 20 * e_3 + 1e-5
    # the training forward (tape, fp32 norm outputs + auto-split) of a SUBSET of the rows equals the sampler's forward bit for bit
    tape = []
    out_t = unet.forward(lat[:2].to(DEV), ts[:2].to(DEV), emb[:2].to(DEV).contiguous(), tape=tape)
    assert torch.equal(out_t, out[:2])
    # one PPO train step against float64 autograd through the oracle
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
    st = sched.set_timesteps(sched.create_state(device=DEV), 50)
    dd = DDIMOracle()
    ost = dd.set_timesteps(dd.create_state(), 50)
    l2, t2, e2, u2 = lat[:2], ts[:2], emb[:2], unc[:2]
    nxt = l2 * 0.95 + 0.05 * torch.randn(l2.shape, generator=g)
    batch = {"latents": l2, "next_latents": nxt, "ts": t2, "log_probs": torch.tensor([-1.2, -0.9]), "advantages": torch.tensor([0.7, -1.1]),
             "prompt_embeds": e2, "uncond_embeds": u2}
    ograds, oinfo, _ = train_step_grads(op, OU.TINY, dd, ost, {k: (v if k == "ts" else v.double()) for k, v in batch.items()}, 5.0, 1.0, 10.0, True, dtype=torch.float64)
    state = AccumulatingTrainState(unet, AdamWConfig())
    state, info = train_step(state, {k: v.to(DEV) for k, v in batch.items()}, st, sched, True, 5.0, 1.0, 10.0, do_opt_update=False)
    gn_o = math.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values()))
    gn = math.sqrt(float((unet.grads.flat.double() ** 2).sum()))
    assert gn == pytest.approx(gn_o, rel=1e-3)
    assert float(info["loss"]) == pytest.approx(oinfo["loss"], rel=1e-3, abs=1e-6)


def test_sampler_ratio_is_one_before_the_first_update_tiny():
    """Sampling, then scoring the stored trajectory with the TRAINING forward of the same weights: log-probs bit-equal -> ratio == 1, approx_kl == 0
    (the contract the PPO clip range of 1e-4 relies on), under f16mx as under bf16x3."""
    from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
    from ddpo_amd.diffusers_patch.pipeline_stable_diffusion import StableDiffusionPipeline
    from ddpo_amd.models.vae import VAEDecoder, VAEConfig
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig, train_step
    from oracle import prng as OP
    op, unet = _tiny(2)
    vae = VAEDecoder(VAEConfig.named("tiny"), DEV)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionPipeline(unet, vae, sched)
    state0 = sched.create_state(device=DEV)
    g = torch.Generator().manual_seed(9)
    emb = torch.randn(4, 77, 64, generator=g).to(DEV)
    neg = torch.randn(1, 77, 64, generator=g).expand(4, -1, -1).contiguous().to(DEV)
    final, lat, nxt, lps, ts = pipe(emb, neg, {"unet": unet.params, "scheduler": state0}, OP.PRNGKey(4), 4, height=64, width=64, guidance_scale=5.0, eta=1.0)
    st = sched.set_timesteps(sched.create_state(device=DEV), 4)
    state = AccumulatingTrainState(unet, AdamWConfig())
    for step in (0, 3):
        batch = {"latents": lat[:2, step].contiguous(), "next_latents": nxt[:2, step].contiguous(), "ts": ts[:2, step].contiguous(),
                 "log_probs": lps[:2, step].contiguous(), "advantages": torch.tensor([0.5, -0.5], device=DEV), "prompt_embeds": emb[:2], "uncond_embeds": neg[:2]}
        state, info = train_step(state, batch, st, sched, True, 5.0, 1.0, 1e-4, do_opt_update=False)
        assert float(info["approx_kl"]) == 0.0 and float(info["clipfrac"]) == 0.0


def test_unet_sd15_single_sample_32x32_accuracy():
    """The real architecture (SD-1.5) on one 32x32 latent: error of the datapath against float64, next to bf16x3 (emulated in round 3: 7e-5 vs 2e-5)."""
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    L.MX_MIN_K = 2560                       # the shipped threshold
    op = OU.init_params(OU.unet_param_shapes(OU.SD15), seed=0)
    unet = UNet2DCondition(UNetConfig.named("sd15"), DEV)
    unet.params.load_dict(op)
    unet.params.pack_bf16(bwd=False)
    g = torch.Generator().manual_seed(0)
    x, ctx = torch.randn(1, 4, 32, 32, generator=g), torch.randn(1, 77, 768, generator=g)
    t = torch.tensor([481], dtype=torch.int32)
    with torch.no_grad():
        ref = OU.unet_forward({k: v.double() for k, v in op.items()}, OU.SD15, x.double(), t, ctx.double())
    out = unet(x.to(DEV), t.to(DEV), ctx.to(DEV))
    with L.datapath("bf16x3"):
        out3 = unet(x.to(DEV), t.to(DEV), ctx.to(DEV))
    rms = lambda a: float((a.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"\n[f16mx SD-1.5 forward, 32x32] rms rel err {rms(out):.2e} (bf16x3 {rms(out3):.2e})")
    assert rms(out) < 2e-4 and rms(out3) < rms(out)


def test_entrypoint_under_f16mx_keeps_ratio_one(tmp_path, monkeypatch):
    """pipeline/policy_gradient.py with DDPO_DATAPATH=f16mx (tiny architecture; MX_MIN_K lowered by the fixture so its convolutions are f16mx layers):
    the first PPO steps of the epoch re-evaluate the sampled trajectory with unchanged weights — approx_kl == 0, clipfrac == 0 — i.e. the training
    forward (fp32 producers, split on the way in, HIP-graph replay) takes the sampler's arithmetic bit for bit through the real entrypoint."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setenv("DDPO_MODEL_CONFIG", "tiny")
    monkeypatch.setenv("DDPO_DATAPATH", "f16mx")
    monkeypatch.setenv("DDPO_ALLOW_SYNTHETIC", "1")
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, root)
    pg = importlib.import_module("pipeline.policy_gradient")
    out = pg.main(["--dataset", "compressed-animals", "--resolution", "64", "--n_inference_steps", "4", "--sample_batch_size", "2", "--train_batch_size", "2",
                   "--num_train_epochs", "1", "--save_freq", "100", "--per_prompt_stats_min_count", "2", "--learning_rate", "1e-4", "--logbase", str(tmp_path / "run")])
    assert L.DATAPATH == "f16mx" and any("mx" in e for e in L.PACKED.values())          # the run really had f16mx layers
    info = np.load(os.path.join(out["localpath"], "train_info/0_0_0.npy"), allow_pickle=True).item()
    assert np.isfinite(info["loss"]).all() and info["approx_kl"].max() < 1e-8 and info["clipfrac"].max() == 0.0
