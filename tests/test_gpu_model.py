"""GPU parity tests of the assembled hot path: U-Net forward, VAE decode and the DDIM sampling loop vs the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ddpo_amd import lib as L
from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
from ddpo_amd.models.vae import VAEDecoder, VAEConfig
from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
from ddpo_amd.diffusers_patch.pipeline_stable_diffusion import StableDiffusionPipeline
from oracle import unet as OU, prng as OP
from oracle.ddim import DDIMOracle
from oracle.sampler import sample as oracle_sample

DEV = "cuda"
SHIPPED = __import__("os").environ.get("DDPO_PARITY_DATAPATH") or L.SHIPPED_DATAPATH      # full-size tests run on what the entrypoints ship (f16mx)


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def tiny():
    op = OU.init_params(OU.unet_param_shapes(OU.TINY), seed=0)
    unet = UNet2DCondition(UNetConfig.named("tiny"), DEV)
    unet.params.load_dict(op)
    ovp = OU.init_params(OU.vae_decoder_param_shapes(OU.VAE_TINY), seed=1)
    vae = VAEDecoder(VAEConfig.named("tiny"), DEV)
    vae.params.load_dict(ovp)
    return op, unet, ovp, vae


@pytest.mark.parametrize("B,hw", [(2, 8), (3, 16), (1, 32)])
def test_unet_tiny_forward(tiny, B, hw):
    op, unet, _, _ = tiny
    g = torch.Generator().manual_seed(B * 100 + hw)
    x = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.tensor([981, 21, 501][:B], dtype=torch.int32)
    ctx = torch.randn(B, 77, 64, generator=g)
    ref = OU.unet_forward({k: v.double() for k, v in op.items()}, OU.TINY, x.double(), t, ctx.double())
    out = unet(x.to(DEV), t.to(DEV), ctx.to(DEV)).cpu()
    assert out.shape == ref.shape
    assert _rel(out.numpy(), ref.numpy()) < 2e-4          # fp32 vs float64 ground truth


def test_vae_tiny_decode(tiny):
    _, _, ovp, vae = tiny
    g = torch.Generator().manual_seed(7)
    z = torch.randn(2, 4, 8, 8, generator=g)
    ref = OU.vae_decode({k: v.double() for k, v in ovp.items()}, OU.VAE_TINY, z.double())
    img = vae.decode(z.to(DEV)).cpu()
    assert img.shape == (2, 64, 64, 3)
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    assert np.abs(img.numpy() - ref.numpy()).max() < 2e-4


def test_sampler_matches_oracle_config1(tiny):
    """BASELINE config 1 geometry: 64x64 px (8x8 latents), 4 DDIM steps, batch 2, guidance 5, eta 1."""
    op, unet, ovp, vae = tiny
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionPipeline(unet, vae, sched)
    state = sched.create_state(device=DEV)
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(2, 77, 64, generator=g)
    neg = torch.randn(1, 77, 64, generator=g).expand(2, -1, -1).contiguous()
    key = OP.PRNGKey(0)
    final, lat, nxt, lps, ts = pipe(emb.to(DEV), neg.to(DEV), {"unet": unet.params, "scheduler": state}, key, 4,
                                    height=64, width=64, guidance_scale=5.0, eta=1.0)
    dd = DDIMOracle()
    ofinal, olat, onxt, olps, ots = oracle_sample(op, OU.TINY, dd, dd.create_state(), emb, neg, key, 4, 64, 64, 5.0, 1.0)
    assert lat.shape == (2, 4, 4, 8, 8) and nxt.shape == (2, 4, 4, 8, 8) and lps.shape == (2, 4) and ts.shape == (2, 4)
    assert np.array_equal(ts.cpu().numpy(), ots)                                   # integer work: bit-exact
    assert torch.equal(lat[:, 1:], nxt[:, :-1]) and torch.equal(final, nxt[:, -1])
    assert _rel(lat[:, 0].cpu().numpy(), olat[:, 0]) < 2e-6                         # initial noise (Threefry)
    assert _rel(final.cpu().numpy(), ofinal) < 1e-3
    assert _rel(nxt.cpu().numpy(), onxt) < 1e-3
    np.testing.assert_allclose(lps.cpu().numpy(), olps, rtol=1e-3, atol=1e-3)
    # jit=True replays a captured HIP graph of the U-Net: bit-identical to the eager launches
    outs_g = pipe(emb.to(DEV), neg.to(DEV), {"unet": unet.params, "scheduler": state}, key, 4, height=64, width=64,
                  guidance_scale=5.0, eta=1.0, jit=True)
    assert torch.equal(outs_g[0], final) and torch.equal(outs_g[3], lps)
    # leading device axis of the reference's pmap convention is accepted and preserved
    outs = pipe(emb[None].to(DEV), neg[None].to(DEV), {"unet": unet.params, "scheduler": state}, key[None], 4,
                height=64, width=64, guidance_scale=5.0, eta=1.0)
    assert outs[0].shape == (1, 2, 4, 8, 8) and torch.equal(outs[0][0], final)
    # reward input: decoded images
    img = vae.decode(final).cpu().numpy()
    oimg = OU.vae_decode(ovp, OU.VAE_TINY, torch.from_numpy(ofinal)).numpy()
    assert np.abs(img - oimg).max() < 5e-3


@pytest.mark.parametrize("datapath", [SHIPPED] + (["fp32"] if __import__("os").environ.get("DDPO_TRAJ_FP32") == "1" else []))
def test_sampler_50_steps_matches_oracle(tiny, datapath):
    """The reference's scan length (num_inference_steps 50, /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:204-270,
    config/base.py) on the toy net: the whole stochastic trajectory, every step's log-prob and the final latents against the oracle
    — error compounding over the full 50 steps on the shipped datapath, graph path (DDPO_TRAJ_FP32=1 adds the exact-fp32 datapath;
    measured round 5, profiles/r05_parity_trajectory_50_steps.log: fp32 2e-6 flat, f16mx 3.6e-5 at step 1 -> 6.6e-5 at step 50, log-probs 1.2e-7)."""
    op, unet, _, _ = tiny
    old = L.DATAPATH
    L.DATAPATH = datapath
    try:
        if datapath != "fp32":
            unet.params.pack_bf16(bwd=False)
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
        pipe = StableDiffusionPipeline(unet, None, sched)
        state = sched.create_state(device=DEV)
        g = torch.Generator().manual_seed(13)
        emb = torch.randn(2, 77, 64, generator=g)
        neg = torch.randn(1, 77, 64, generator=g).expand(2, -1, -1).contiguous()
        key = OP.PRNGKey(5)
        T = 50
        final, lat, nxt, lps, ts = pipe(emb.to(DEV), neg.to(DEV), {"unet": unet.params, "scheduler": state}, key, T,
                                        height=128, width=128, guidance_scale=5.0, eta=1.0, jit=True)
        dd = DDIMOracle()
        ofinal, olat, onxt, olps, ots = oracle_sample(op, OU.TINY, dd, dd.create_state(), emb, neg, key, T, 128, 128, 5.0, 1.0)
        assert np.array_equal(ts.cpu().numpy(), ots)
        e_step = [_rel(nxt[:, i].cpu().numpy(), onxt[:, i]) for i in range(T)]
        e_lp = float(np.abs(lps.cpu().numpy() - olps).max() / np.abs(olps).max())
        from conftest import parity_record
        parity_record(f"[tiny sampler, 50 steps, graph path] {datapath}: latent rel err at steps 1/10/25/50 "
                      f"{e_step[0]:.1e} {e_step[9]:.1e} {e_step[24]:.1e} {e_step[49]:.1e} (max {max(e_step):.1e})  log-probs rel {e_lp:.1e}")
        assert max(e_step) < 1e-3 and e_lp < 1e-3 and _rel(final.cpu().numpy(), ofinal) < 1e-3
    finally:
        L.DATAPATH = old
        L.PACKED.clear()


def test_time_projection_table_is_bit_identical_to_per_step_time_path(tiny, monkeypatch):
    """precompute_timesteps / select_timestep: the sampler runs the time path (embedding MLP + every ResBlock's time_emb_proj) once
    per call for all T steps; trajectories and log-probs must not change by a bit, eagerly and under HIP-graph replay."""
    op, unet, ovp, vae = tiny
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionPipeline(unet, vae, sched)
    state = sched.create_state(device=DEV)
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(2, 77, 64, generator=g).to(DEV)
    neg = torch.randn(1, 77, 64, generator=g).expand(2, -1, -1).contiguous().to(DEV)
    key = OP.PRNGKey(11)
    run = lambda jit: pipe(emb, neg, {"unet": unet.params, "scheduler": state}, key, 5, height=64, width=64, guidance_scale=5.0,
                           eta=1.0, jit=jit)
    monkeypatch.setenv("DDPO_TEMB_CACHE", "0")
    ref = [t.clone() for t in run(False)]
    monkeypatch.setenv("DDPO_TEMB_CACHE", "1")
    for jit in (False, True, True):                      # second jit call replays the graph captured by the first
        out = run(jit)
        assert all(torch.equal(a, b) for a, b in zip(out, ref)), jit
    assert unet._temb is not None and not unet._temb_active and unet._temb["table"].shape[0] == 5
    # outside a sampling call forward() still computes the time path itself (training, per-sample timesteps)
    x = torch.randn(3, 4, 8, 8, device=DEV)
    t = torch.tensor([981, 21, 501], dtype=torch.int32, device=DEV)
    c = torch.randn(3, 77, 64, device=DEV)
    y0 = unet(x, t, c)
    unet.precompute_timesteps([981, 21, 501])
    unet.release_timesteps()
    assert torch.equal(unet(x, t, c), y0)


@pytest.mark.parametrize("datapath", ["bf16x3", "f16mx"])
def test_context_kv_images_are_bit_identical_to_per_step_staging(monkeypatch, datapath):
    """precompute_context also packs the text K / V once into the attention kernels' per-tile images; every cross-attention of the
    sampling call then runs from them (ddpo_attention_fwd_bf16x3_images) instead of splitting / transposing K and V in every query tile
    of every step.  Same kernels' arithmetic: the U-Net output must not change by a bit (bf16x3 datapath; fp32 keeps k, v).
    Own model instance (head dim 16 at every level): the module fixture's captured graphs must keep their context buffers."""
    monkeypatch.setattr(L, "DATAPATH", datapath)             # f16mx: the f16p attention operator and its images (V as f16 hi / lo + ones row)
    unet = UNet2DCondition(UNetConfig.named("tiny21"), DEV)
    unet.params.init_synthetic(3)
    unet.params.pack_bf16()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 4, 8, 8, generator=g).to(DEV)
    t = torch.tensor([481, 481, 481, 481], dtype=torch.int32, device=DEV)
    c = torch.randn(4, 77, 96, generator=g).to(DEV)
    y0 = unet(x, t, c).clone()
    unet.precompute_context(c)
    try:
        assert unet._ctx_kv and all(ent[2] is not None for ent in unet._ctx_kv.values())
        y1 = unet(x, t, c).clone()
    finally:
        unet.release_context()
    monkeypatch.setenv("DDPO_CTX_IMAGES", "0")
    unet._ctx_kv.clear()
    unet.precompute_context(c)
    try:
        assert all(ent[2] is None for ent in unet._ctx_kv.values())
        y2 = unet(x, t, c).clone()
    finally:
        unet.release_context()
    assert torch.equal(y0, y1) and torch.equal(y0, y2)


def test_graph_replay_survives_a_change_of_sampling_geometry(tiny):
    """the text-context K/V buffers are keyed by (layer, context rows) and never replaced, so a HIP graph captured
    for batch 2 still reads valid K/V after the same U-Net sampled batch 1 in between (it used to replay against freed memory)."""
    _, unet, _, vae = tiny
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionPipeline(unet, vae, sched)
    state = sched.create_state(device=DEV)
    g = torch.Generator().manual_seed(11)
    emb2 = torch.randn(2, 77, 64, generator=g).to(DEV)
    neg2 = torch.randn(1, 77, 64, generator=g).expand(2, -1, -1).contiguous().to(DEV)
    emb1 = torch.randn(1, 77, 64, generator=g).to(DEV)
    key = OP.PRNGKey(5)
    run = lambda e, n, jit: pipe(e, n, {"unet": unet.params, "scheduler": state}, key, 4, height=64, width=64, guidance_scale=5.0,
                                 eta=1.0, jit=jit)
    eager2 = run(emb2, neg2, False)
    first2 = run(emb2, neg2, True)                      # captures the batch-2 graph
    run(emb1, neg2[:1].contiguous(), True)              # another geometry: new K/V buffers, its own graph
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]     # recycle anything that was freed
    again2 = run(emb2, neg2, True)                      # replay of the batch-2 graph
    del junk
    for a, b, c in zip(eager2, first2, again2):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert len({k[1] for k in unet._ctx_kv}) == 2       # both context geometries stay resident


def test_unet_sd15_single_sample_64x64():
    """Full SD-1.x architecture (859.5 M params) at the 512^2 latent size, one sample, against the torch-CPU oracle."""
    shapes = OU.unet_param_shapes(OU.SD15)
    op = OU.init_params(shapes, seed=0)
    unet = UNet2DCondition(UNetConfig.named("sd15"), DEV)
    assert unet.params.n_params == 859520964
    unet.params.load_dict(op)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 4, 64, 64, generator=g)
    t = torch.tensor([481], dtype=torch.int32)
    ctx = torch.randn(1, 77, 768, generator=g)
    with torch.no_grad():
        ref = OU.unet_forward(op, OU.SD15, x, t, ctx)
    out = unet(x.to(DEV), t.to(DEV), ctx.to(DEV)).cpu()
    assert _rel(out.numpy(), ref.numpy()) < 1e-3


def test_unet_sd21_single_sample_96x96_shipped_datapath():
    """BASELINE configs[4] architecture at full size: SD-2.1 U-Net (865.9 M params; linear proj_in/out, d_head 64 with
    5/10/20/20 heads, 1024-wide context) on a 768^2 latent (96x96 -> self-attention over 9216 keys), one sample, on the
    shipped datapath (buffer-addressed GEMMs, f16mx long reductions, pre-packed K/V attention) against the torch-CPU fp32 oracle."""
    shapes = OU.unet_param_shapes(OU.SD21)
    op = OU.init_params(shapes, seed=0)
    old = L.DATAPATH
    L.DATAPATH = SHIPPED
    try:
        unet = UNet2DCondition(UNetConfig.named("sd21"), DEV)
        assert unet.params.n_params == 865910724
        unet.params.load_dict(op)
        unet.params.pack_bf16(bwd=False)
        g = torch.Generator().manual_seed(21)
        x = torch.randn(1, 4, 96, 96, generator=g)
        t = torch.tensor([261], dtype=torch.int32)
        ctx = torch.randn(1, 77, 1024, generator=g)
        with torch.no_grad():
            ref = OU.unet_forward(op, OU.SD21, x, t, ctx)
        out = unet(x.to(DEV), t.to(DEV), ctx.to(DEV)).cpu()
        assert out.shape == ref.shape == (1, 4, 96, 96)
        assert _rel(out.numpy(), ref.numpy()) < 1e-3
    finally:
        L.DATAPATH = old
        L.PACKED.clear()


@pytest.mark.timeout(900)
def test_sampler_sd21_full_size_96x96_graph_path():
    """BASELINE configs[4] (C5) sampler at size: the full SD-2.1 U-Net, 96x96 latents, classifier-free guidance, v-prediction DDIM
    with eta = 1, THREE denoising steps through the captured-HIP-graph path (`jit=True`, the reference's only live path:
    /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:355-365), time-projection table and cached text K/V
    included, against the fp32 oracle's sampling loop: timesteps equal, trajectories / final latents / log-probs within the
    north-star tolerance.  The eager path must agree with the graph replay bit for bit."""
    op = OU.init_params(OU.unet_param_shapes(OU.SD21), seed=0)
    old = L.DATAPATH
    L.DATAPATH = SHIPPED
    try:
        unet = UNet2DCondition(UNetConfig.named("sd21"), DEV)
        unet.params.load_dict(op)
        unet.params.pack_bf16(bwd=False)
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False,
                              steps_offset=1, prediction_type="v_prediction")
        pipe = StableDiffusionPipeline(unet, None, sched)
        state = sched.create_state(device=DEV)
        g = torch.Generator().manual_seed(31)
        emb = torch.randn(1, 77, 1024, generator=g)
        neg = torch.randn(1, 77, 1024, generator=g)
        key = OP.PRNGKey(11)
        T = 3
        args = (emb.to(DEV), neg.to(DEV), {"unet": unet.params, "scheduler": state}, key, T)
        final, lat, nxt, lps, ts = pipe(*args, height=768, width=768, guidance_scale=5.0, eta=1.0, jit=True)
        final_e, lat_e, nxt_e, lps_e, ts_e = pipe(*args, height=768, width=768, guidance_scale=5.0, eta=1.0, jit=False)
        assert torch.equal(final, final_e) and torch.equal(nxt, nxt_e) and torch.equal(lps, lps_e)
        dd = DDIMOracle(prediction_type="v_prediction")
        with torch.no_grad():
            ofinal, olat, onxt, olps, ots = oracle_sample(op, OU.SD21, dd, dd.create_state(), emb, neg, key, T, 768, 768, 5.0, 1.0)
        assert final.shape == (1, 4, 96, 96) and np.array_equal(ts.cpu().numpy(), ots)
        e_f, e_n = _rel(final.cpu().numpy(), ofinal), _rel(nxt.cpu().numpy(), onxt)
        e_lp = float(np.abs(lps.cpu().numpy() - olps).max() / np.abs(olps).max())
        from conftest import parity_record
        parity_record(f"\n[sd21 96x96 sampler, {T} steps, graph path] {SHIPPED}: final latents {e_f:.2e}  trajectory {e_n:.2e}  log-probs rel {e_lp:.2e}")
        assert e_f < 1e-3 and e_n < 1e-3 and e_lp < 1e-3
    finally:
        L.DATAPATH = old
        L.PACKED.clear()


@pytest.mark.parametrize("datapath", ["fp32", "bf16x3"])
def test_sd21_shaped_config_sampler_and_train_step(datapath):
    """BASELINE configs[4] shape class on the toy scale: SD-2.1 architecture switches (linear proj_in/out, per-level head
    counts with d_head 16, 96-wide text context) + v-prediction DDIM, sampler and one PPO step against the oracle."""
    import math
    from ddpo_amd import lib as L
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig, train_step
    from oracle.sampler import train_step_grads
    old = L.DATAPATH
    L.DATAPATH = datapath
    try:
        op = OU.init_params(OU.unet_param_shapes(OU.TINY21), seed=4)
        unet = UNet2DCondition(UNetConfig.named("tiny21"), DEV)
        unet.params.load_dict(op)
        if datapath != "fp32":
            unet.params.pack_bf16()
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False,
                              steps_offset=1, prediction_type="v_prediction")
        pipe = StableDiffusionPipeline(unet, None, sched)
        state = sched.create_state(device=DEV)
        g = torch.Generator().manual_seed(9)
        emb = torch.randn(2, 77, 96, generator=g)
        neg = torch.randn(1, 77, 96, generator=g).expand(2, -1, -1).contiguous()
        key = OP.PRNGKey(7)
        final, lat, nxt, lps, ts = pipe(emb.to(DEV), neg.to(DEV), {"unet": unet.params, "scheduler": state}, key, 4,
                                        height=128, width=128, guidance_scale=5.0, eta=1.0)
        dd = DDIMOracle(prediction_type="v_prediction")
        ofinal, olat, onxt, olps, ots = oracle_sample(op, OU.TINY21, dd, dd.create_state(), emb, neg, key, 4, 128, 128, 5.0, 1.0)
        assert np.array_equal(ts.cpu().numpy(), ots)
        assert _rel(final.cpu().numpy(), ofinal) < 1e-3
        np.testing.assert_allclose(lps.cpu().numpy(), olps, rtol=1e-3, atol=1e-3)
        # one PPO micro-step on the sampled trajectory at timestep index 1
        ost = dd.set_timesteps(dd.create_state(), 4)
        st4 = sched.set_timesteps(state, 4)
        batch = {"latents": torch.from_numpy(olat[:, 1]), "next_latents": torch.from_numpy(onxt[:, 1]), "ts": torch.from_numpy(ots[:, 1].copy()),
                 "log_probs": torch.from_numpy(olps[:, 1]) + torch.tensor([3e-5, -2e-5]), "advantages": torch.tensor([0.9, -1.4]),
                 "prompt_embeds": emb, "uncond_embeds": neg}
        ograds, oinfo, _ = train_step_grads(op, OU.TINY21, dd, ost, {k: (v if k == "ts" else v.double()) for k, v in batch.items()},
                                            5.0, 1.0, 1e-4, True, dtype=torch.float64)
        tstate = AccumulatingTrainState(unet, AdamWConfig())
        tstate, info = train_step(tstate, {k: v.to(DEV) for k, v in batch.items()}, st4, sched, True, 5.0, 1.0, 1e-4, do_opt_update=False)
        gn_o = math.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values()))
        gn = math.sqrt(float((unet.grads.flat.double() ** 2).sum()))
        print(f"\n[tiny21 T=4 train step] {datapath}: grad-norm rel err {abs(gn - gn_o) / gn_o:.2e}")
        assert gn == pytest.approx(gn_o, rel=1e-3)       # north_star tolerance on BOTH datapaths, at the reference's clip range 1e-4
    finally:
        L.DATAPATH = old
        L.PACKED.clear()


@pytest.mark.parametrize("family,ctx_dim,datapath", [("tiny", 64, "fp32"), ("tiny", 64, "bf16x3"), ("tiny21", 96, "bf16x3"), ("tiny", 64, "f16mx")])
def test_skip_concat_in_place_is_bit_identical_to_copies(family, ctx_dim, datapath, monkeypatch):
    """Round 4: the sampling forward forms every skip concatenation in place — the down path's producers write their output into its column
    range of the consuming up block's concat buffer, the up path's producers into the columns in front of it — instead of two ddpo_copy_cols
    launches per up block.  Only row strides change: outputs (and the planes the sampler convolutions read) must not move by a bit, with and
    without the CFG-shared front of the network, eagerly and under graph replay."""
    from ddpo_amd.models import unet as U
    monkeypatch.setattr(L, "DATAPATH", datapath)
    if datapath == "f16mx":
        monkeypatch.setattr(L, "MX_MIN_K", 256)
    unet = UNet2DCondition(UNetConfig.named(family), DEV)
    unet.params.init_synthetic(2)
    if datapath != "fp32":
        unet.params.pack_bf16(bwd=False)
    g = torch.Generator().manual_seed(12)
    x1 = torch.randn(3, 4, 16, 16, generator=g).to(DEV)
    x = torch.cat([x1, x1])
    t = torch.full((6,), 481, dtype=torch.int32, device=DEV)
    c = torch.randn(6, 77, ctx_dim, generator=g).to(DEV)
    out = {}
    for inplace in (False, True):
        monkeypatch.setattr(U, "SKIP_INPLACE", inplace)
        before = L.gemm_tile_launch_counts()
        out[inplace] = [unet(x, t, c).clone(), unet(x, t, c, cfg_dup=True).clone(), unet.forward_graphed(x, t, c, cfg_dup=True).clone()]
        unet._graphs.clear()
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)
    assert torch.equal(out[True][0], out[True][1]) and torch.equal(out[True][1], out[True][2])
    # the training forward keeps contiguous tensors and the same bits
    monkeypatch.setattr(U, "SKIP_INPLACE", True)
    assert torch.equal(unet.forward(x, t, c, tape=[]), out[True][0])


@pytest.mark.parametrize("family,ctx_dim,datapath,planes_all", [("tiny", 64, "bf16x3", False), ("tiny21", 96, "bf16x3", True), ("tiny", 64, "f16mx", True)])
def test_fused_qkv_projection_of_the_sampling_self_attention_is_bit_identical(family, ctx_dim, datapath, planes_all, monkeypatch):
    """Round 5: the sampling forward projects q, k, v of a self-attention in ONE launch on the (K, 3C) concatenation of the three kernels
    (ParamStore.pack_bf16 keeps it next to them) and hands the attention column slices of the result (row stride 3C).  Every column
    accumulates in the same k order as in its own projection, so the U-Net output must not move by a bit — fp32-fed and plane-fed, eagerly and
    under graph replay — and the fused path must actually have been taken (strided q / k / v reach the attention entry)."""
    from ddpo_amd.models import unet as U
    monkeypatch.setattr(L, "DATAPATH", datapath)
    monkeypatch.setattr(L, "PLANES_ALL", planes_all)
    if datapath == "f16mx":
        monkeypatch.setattr(L, "MX_MIN_K", 256)
    unet = UNet2DCondition(UNetConfig.named(family), DEV)
    unet.params.init_synthetic(4)
    unet.params.pack_bf16(bwd=False)
    assert len(unet.params.fused_qkv) == len([n for n in unet.params.views if n.endswith(".attn1.to_q.kernel")]) > 0
    g = torch.Generator().manual_seed(17)
    x = torch.randn(4, 4, 16, 16, generator=g).to(DEV)
    t = torch.tensor([981, 21, 501, 481], dtype=torch.int32, device=DEV)
    c = torch.randn(4, 77, ctx_dim, generator=g).to(DEV)
    strided = {"n": 0}
    real_attn = L.attention

    def attn(*a, **kw):
        strided["n"] += bool(kw.get("ldq"))
        return real_attn(*a, **kw)

    monkeypatch.setattr(L, "attention", attn)
    out = {}
    for on in (False, True):
        monkeypatch.setattr(U, "QKV_FUSED", on)
        out[on] = unet(x, t, c).clone()
    assert strided["n"] == len(unet.params.fused_qkv)            # every self-attention of the fused forward, none of the unfused one
    assert torch.equal(out[True], out[False])
    unet.params["down_blocks_0.attentions_0.transformer_blocks_0.attn1.to_k.kernel"].mul_(1.5)      # an "optimizer update": re-packing refreshes the copy
    unet.params.pack_bf16(bwd=False)
    o2 = unet(x, t, c)
    monkeypatch.setattr(U, "QKV_FUSED", False)
    assert torch.equal(o2, unet(x, t, c)) and not torch.equal(o2, out[True])


@pytest.mark.parametrize("family,ctx_dim,datapath", [("tiny", 64, "bf16x3"), ("tiny21", 96, "bf16x3"), ("tiny", 64, "f16mx")])
def test_plane_handover_behind_attention_and_ff2_is_bit_identical(family, ctx_dim, datapath, monkeypatch):
    """Round 4: in the sampling forward the attention kernels hand their result to to_out, and the second feed-forward GEMM hands h3 to
    proj_out, as bf16 hi / lo planes wherever the consumer is faster plane-fed (lib.planes_pay: the 64x64 level of SD) — no fp32 tensor in
    between.  The planes are the exact split the fp32-fed loader applies and the plane-fed kernels are bit-identical to the fp32-fed ones, so
    the U-Net output must not move by a bit (DDPO_PLANES_ALL semantics make the tiny models take the path), eagerly, with the cached
    text-context images and under graph replay; and the path must actually have been taken."""
    from ddpo_amd.models import unet as U
    monkeypatch.setattr(L, "DATAPATH", datapath)
    monkeypatch.setattr(L, "PLANES_ALL", True)
    if datapath == "f16mx":
        monkeypatch.setattr(L, "MX_MIN_K", 256)
    unet = UNet2DCondition(UNetConfig.named(family), DEV)
    unet.params.init_synthetic(2)
    unet.params.pack_bf16(bwd=False)
    g = torch.Generator().manual_seed(13)
    x1 = torch.randn(3, 4, 16, 16, generator=g).to(DEV)
    x = torch.cat([x1, x1])
    t = torch.full((6,), 481, dtype=torch.int32, device=DEV)
    c = torch.randn(6, 77, ctx_dim, generator=g).to(DEV)
    seen = {"attn": 0, "img": 0, "h3": 0}
    real_attn, real_img, real_gc = L.attention, L.attention_from_images, L.gemm_conv

    def attn(*a, **kw):
        seen["attn"] += bool(kw.get("planes_out"))
        return real_attn(*a, **kw)

    def img(*a, **kw):
        seen["img"] += bool(kw.get("planes_out"))
        return real_img(*a, **kw)

    def gc(*a, **kw):
        seen["h3"] += kw.get("planes_out") == "only"
        return real_gc(*a, **kw)

    monkeypatch.setattr(L, "attention", attn)
    monkeypatch.setattr(L, "attention_from_images", img)
    monkeypatch.setattr(L, "gemm_conv", gc)
    out = {}
    for on in (False, True):
        monkeypatch.setattr(U, "ATTN_PLANES", on)
        monkeypatch.setattr(U, "H3_PLANES", on)
        res = [unet(x, t, c).clone(), unet(x, t, c, cfg_dup=True).clone()]
        unet.precompute_context(c)                           # cross-attention from the packed text-context images
        try:
            res.append(unet(x, t, c).clone())
            res.append(unet.forward_graphed(x, t, c, cfg_dup=True).clone())
        finally:
            unet.release_context()
        unet._graphs.clear()
        out[on] = res
        if not on:
            assert seen == {"attn": 0, "img": 0, "h3": 0}
    assert seen["attn"] > 0 and seen["img"] > 0 and seen["h3"] > 0, seen
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)
    # the training forward never takes the hand-over (its backward reads the fp32 tensors) and keeps the same bits
    n = dict(seen)
    assert torch.equal(unet.forward(x, t, c, tape=[]), out[True][0]) and seen == n


@pytest.mark.parametrize("M", [1024, 2048, 4096])
def test_fused_qkv_projection_at_sd15_mid_block_shapes(M, monkeypatch):
    """the fused (K, 3C) projection keeps every column's k ORDER, but the tile / split-K choice depends on the column count —
    at M = 1024, K = N = 1280 the separate projections run 128 x 64 tiles with the reduction split in three while N = 3840 runs unsplit, so
    the fp32 summation order differs there.  Hold the fused launch to the three separate ones at the 16x16-level shapes of SD-1.5
    (C = 1280; M = 4096 is the sampling batch of 16, where both are unsplit and the bits must agree)."""
    monkeypatch.setattr(L, "DATAPATH", "bf16x3")
    torch.manual_seed(11)
    C = 1280
    x = torch.randn(M, C, device=DEV)
    ws = [(torch.randn(C, C, device=DEV) / C ** 0.5).contiguous() for _ in range(3)]
    fused = torch.cat(ws, dim=1).contiguous()
    try:
        for w in ws + [fused]:
            L.pack_weights(w, bwd=False)
        sep = torch.cat([L.linear(x, w) for w in ws], dim=1)
        one = L.linear(x, fused)
        torch.cuda.synchronize()
        ref = x.double() @ fused.double()
        scale = float(ref.abs().max())
        assert float((one - sep).abs().max()) <= 2e-6 * scale              # summation order only
        assert float((one.double() - ref).abs().max()) < 1e-4 * scale and float((sep.double() - ref).abs().max()) < 1e-4 * scale
        if M == 4096:
            assert torch.equal(one, sep)
    finally:
        L.PACKED.clear()
