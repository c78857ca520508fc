"""The bench's own geometry against the oracle in ONE test.

BASELINE configs[1] as `bench.py` runs it — SD-1.5, 64x64 latents (512^2 px), `sample_batch_size 8` (U-Net batch 16 under classifier-free
guidance), `jit=True` (captured HIP graph), the shipped datapath (lib.SHIPPED_DATAPATH), cfg_dup, time-projection table, cached text-context
K / V images — for FIVE DDIM steps (DDPO_HEADLINE_STEPS; two until round 6), against `oracle.sampler.sample` on the same prompts / key:
/root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:204-270.

Every other oracle comparison at size runs b = 1, where the 256x320 "tall" GEMM tile (16 % of the sampling step) is never selected
(`ntall >= 200` needs a U-Net batch >= 13 at 64x64); it was validated only transitively (bit-identity to the 128-row kernels through
tools/native/kernel_probe).  Here the tall tile, the 128x320 f16mx tile, split-K and the LDS-DMA attention all run at the headline shapes
and the launch counters of the library (ddpo_gemm_tile_launch_counts) prove it.

Host time: 5 x the oracle U-Net on a batch of 16 at 64x64 (about 3 minutes on the GPU box's cores; VERDICT r05 weak 1b: the tall tiles
must see a trajectory, not two steps)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ddpo_amd import lib as L
from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
from ddpo_amd.diffusers_patch.pipeline_stable_diffusion import StableDiffusionPipeline
from oracle import unet as OU, prng as OP
from oracle.ddim import DDIMOracle
from oracle.sampler import sample as oracle_sample

DEV = "cuda"


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.timeout(1500)
def test_headline_geometry_sampler_matches_oracle_and_runs_the_tall_tile():
    datapath = os.environ.get("DDPO_PARITY_DATAPATH") or L.SHIPPED_DATAPATH
    old = L.DATAPATH
    L.DATAPATH = datapath
    try:
        op = OU.init_params(OU.unet_param_shapes(OU.SD15), seed=0)
        unet = UNet2DCondition(UNetConfig.named("sd15"), DEV)
        unet.params.load_dict(op)
        unet.params.pack_bf16(bwd=False)
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
        pipe = StableDiffusionPipeline(unet, None, sched)
        state = sched.create_state(device=DEV)
        B, T = 8, int(os.environ.get("DDPO_HEADLINE_STEPS", "5"))
        g = torch.Generator().manual_seed(41)
        emb = torch.randn(B, 77, 768, generator=g)
        neg = torch.randn(1, 77, 768, generator=g).expand(B, -1, -1).contiguous()
        key = OP.PRNGKey(17)
        before = L.gemm_tile_launch_counts()
        args = (emb.to(DEV), neg.to(DEV), {"unet": unet.params, "scheduler": state}, key, T)
        final, lat, nxt, lps, ts = pipe(*args, height=512, width=512, guidance_scale=5.0, eta=1.0, jit=True)
        torch.cuda.synchronize()
        after = L.gemm_tile_launch_counts()
        ran = {k: after[k] - before[k] for k in after}
        print(f"\n[headline geometry] datapath {datapath}; GEMM launches by tile class (capture + warm-up forwards): {ran}")
        # the headline configuration's tiles: the tall tile needs the U-Net batch of 16; f16mx layers exist iff the datapath is f16mx
        assert ran["tall_256x320"] > 0, ran
        assert ran["wide_128x320"] > 0 and ran["t128x128"] > 0 and ran["t128x64"] > 0 and ran["splitk_reduce"] > 0, ran
        assert (ran["f16mx"] > 0) == (datapath == "f16mx"), ran
        assert ran["generic_loader"] <= 3 * 2, ran            # conv_in / conv_out only (Cin = 4 / N = 4), per forward of warm-up + capture
        # graph replay == eager launches, bit for bit, at this geometry too
        final_e, lat_e, nxt_e, lps_e, ts_e = pipe(*args, height=512, width=512, guidance_scale=5.0, eta=1.0, jit=False)
        assert torch.equal(final, final_e) and torch.equal(nxt, nxt_e) and torch.equal(lps, lps_e)

        dd = DDIMOracle()
        with torch.no_grad():
            ofinal, olat, onxt, olps, ots = oracle_sample(op, OU.SD15, dd, dd.create_state(), emb, neg, key, T, 512, 512, 5.0, 1.0)
        assert final.shape == (B, 4, 64, 64) and lat.shape == (B, T, 4, 64, 64) and lps.shape == (B, T)
        assert np.array_equal(ts.cpu().numpy(), ots)                                    # integer work: bit-exact
        assert torch.equal(lat[:, 1:], nxt[:, :-1]) and torch.equal(final, nxt[:, -1])
        e0 = _rel(lat[:, 0].cpu().numpy(), olat[:, 0])                                  # initial noise (Threefry + ErfInv)
        e_f, e_n = _rel(final.cpu().numpy(), ofinal), _rel(nxt.cpu().numpy(), onxt)
        e_lp = float(np.abs(lps.cpu().numpy() - olps).max() / np.abs(olps).max())
        per_sample = [_rel(final[i].cpu().numpy(), ofinal[i]) for i in range(B)]
        from conftest import parity_record
        parity_record(f"[headline geometry] {datapath} B={B}, {T} steps, graph path: initial noise {e0:.2e}  final latents {e_f:.2e}  trajectory {e_n:.2e}  "
                      f"log-probs rel {e_lp:.2e}  per-sample final {', '.join(f'{v:.1e}' for v in per_sample)}")
        assert e0 < 2e-6
        assert e_f < 1e-3 and e_n < 1e-3 and e_lp < 1e-3                               # north_star tolerance
    finally:
        L.DATAPATH = old
        L.PACKED.clear()


@pytest.mark.timeout(1500)
def test_headline_size_trajectory_error_growth_over_many_steps():
    """The reference's sampler is a 50-iteration scan (/root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:204-270)
    and BASELINE's metric is quoted at 50 steps: hold a LONG stochastic trajectory of the full SD-1.5 U-Net at 64x64 latents, B = 1,
    on the shipped datapath and the captured-graph path to the oracle's sampling loop and record how the error grows step by step
    (VERDICT r04 missing 2 / ADVICE r04: "at least 20 steps at the headline geometry").  `num_inference_steps = 50` sets the
    timestep grid of the headline run; DDPO_TRAJ_STEPS (default 50 since round 6: ALL of them, so that the driver's suite sees what the
    builder's log claims — VERDICT r05 weak 1b; the oracle costs ~4.4 s of host time per CFG step) is how many of its 50 steps are walked."""
    datapath = os.environ.get("DDPO_PARITY_DATAPATH") or L.SHIPPED_DATAPATH
    n_walk = int(os.environ.get("DDPO_TRAJ_STEPS", "50"))
    old = L.DATAPATH
    L.DATAPATH = datapath
    try:
        op = OU.init_params(OU.unet_param_shapes(OU.SD15), seed=0)
        unet = UNet2DCondition(UNetConfig.named("sd15"), DEV)
        unet.params.load_dict(op)
        unet.params.pack_bf16(bwd=False)
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
        pipe = StableDiffusionPipeline(unet, None, sched)
        state = sched.create_state(device=DEV)
        B, T = 1, 50
        g = torch.Generator().manual_seed(43)
        emb = torch.randn(B, 77, 768, generator=g)
        neg = torch.randn(1, 77, 768, generator=g)
        key = OP.PRNGKey(23)
        final, lat, nxt, lps, ts = pipe(emb.to(DEV), neg.to(DEV), {"unet": unet.params, "scheduler": state}, key, T,
                                        height=512, width=512, guidance_scale=5.0, eta=1.0, jit=True)
        torch.cuda.synchronize()
        dd = DDIMOracle()

        class _Stop(Exception):
            pass

        walked = []

        def unet_fn(x, t, c):                        # the oracle U-Net, stopping the oracle's loop after n_walk steps
            if len(walked) == n_walk:
                raise _Stop
            walked.append(int(t[0]))
            with torch.no_grad():
                return OU.unet_forward(op, OU.SD15, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(c)).numpy()

        rec = {}
        orig_step = dd.step

        def step(st, eps, t, x, noise=None, eta=0.0):
            new, lp = orig_step(st, eps, t, x, noise=noise, eta=eta)
            rec.setdefault("nxt", []).append(new); rec.setdefault("lp", []).append(lp)
            return new, lp

        dd.step = step
        try:
            oracle_sample(op, OU.SD15, dd, dd.create_state(), emb, neg, key, T, 512, 512, 5.0, 1.0, unet_fn=unet_fn)
        except _Stop:
            pass
        n = len(rec["nxt"])
        assert n == min(n_walk, T) and walked == [int(v) for v in ts[0, :n].cpu().numpy()]          # integer work: bit-exact
        onxt, olp = np.stack(rec["nxt"], 1), np.stack(rec["lp"], 1)
        e_step = [_rel(nxt[:, i].cpu().numpy(), onxt[:, i]) for i in range(n)]
        lp_scale = float(np.abs(olp).max())
        e_lp = [float(np.abs(lps[:, i].cpu().numpy() - olp[:, i]).max() / lp_scale) for i in range(n)]
        from conftest import parity_record
        parity_record(f"[headline size, long trajectory] {datapath} B={B}, {n} of {T} steps, graph path: latent rel err per step "
                      f"{' '.join(f'{v:.1e}' for v in e_step)} | log-prob rel err per step {' '.join(f'{v:.1e}' for v in e_lp)}")
        assert max(e_step) < 1e-3 and max(e_lp) < 1e-3                                   # north_star tolerance, at EVERY step
    finally:
        L.DATAPATH = old
        L.PACKED.clear()
