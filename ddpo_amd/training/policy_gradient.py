"""PPO-clip policy-gradient step on stored DDIM log-probs + gradient-accumulating train state.

Host-side mirror of /root/reference/ddpo/training/policy_gradient.py:
  AccumulatingTrainState :13-57, ADV_CLIP_MAX :60, train_step :63-146
and of the optimizer the entrypoint builds (/root/reference/pipeline/policy_gradient.py:130-150):
  optax.chain(clip_by_global_norm(max_grad_norm), adamw(lr, b1, b2, eps, weight_decay, mu_dtype=bf16)).

MI355X design notes
  * cond and uncond U-Net passes of `train_cfg` are run as ONE batch-2b pass ([uncond; cond] contexts, the same
    latents twice) — identical arithmetic per sample, half the launches, twice the rows per GEMM.
  * gradients are accumulated in place in one flat fp32 buffer by the wgrad kernels (grad_acc += g is free);
    `lax.pmean(grad)` of the reference (:141, every micro-step) becomes ONE RCCL all-reduce of that buffer per
    optimizer update — mean-over-ranks and sum-over-steps commute.
  * the update itself is one fused kernel: 1/(n_acc+1) scaling, global-norm clip, AdamW with bf16 first moment.
"""
import numpy as np
import torch

from .. import lib as L

ADV_CLIP_MAX = 10.0
# PPO micro-steps per U-Net forward/backward in the entrypoint's training loop (train_steps_fused).  16 turns the U-Net batch of 4
# (2 samples x CFG) into 64 rows-of-latents = 1024 of the 256x320 GEMM tiles at the 64x64 level = exactly 4 rounds of the 256 CUs;
# round 1's default of 10 (batch 40 = 2.5 rounds) sat in a quantisation hole: 42.4 (k = 10) vs 44.1 (k = 8) vs 45.1 (k = 16)
# sample-timesteps/s on one box (profiles/r02_ab_train_fuse.log; k = 1: 28).  The default 50 timesteps run as 16 + 16 + 16 + 2.
# DDPO_TRAIN_FUSE=1 restores one launch per micro-step.
DEFAULT_TRAIN_FUSE = 16


def train_fuse_default(unet_rows_per_micro_step=None, latent_pixels=None):
    """Micro-steps per launch.  An explicit DDPO_TRAIN_FUSE is taken as is; the default of 16 is reduced for geometries whose
    activation tape would outgrow the footprint run on hardware (U-Net batch 64 at 64x64 latents = 262,144 latent pixels per
    launch): SD-2.1 at 96x96 latents fuses 7 micro-steps."""
    import os
    if "DDPO_TRAIN_FUSE" in os.environ:
        return max(1, int(os.environ["DDPO_TRAIN_FUSE"]))
    k = DEFAULT_TRAIN_FUSE
    if unet_rows_per_micro_step and latent_pixels:
        k = min(k, max(1, (64 * 4096) // (int(unet_rows_per_micro_step) * int(latent_pixels))))
    return k


class AdamWConfig:
    """Hyper-parameters of pipeline/policy_gradient.py:130-150 (defaults = config/base.py `pg`)."""

    def __init__(self, learning_rate=1e-5, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0,
                 mu_decay_in_bf16=True):
        self.learning_rate, self.b1, self.b2, self.eps = learning_rate, b1, b2, eps
        self.weight_decay, self.max_grad_norm = weight_decay, max_grad_norm
        self.mu_decay_in_bf16 = mu_decay_in_bf16


class AccumulatingTrainState:
    """TrainState that accumulates gradients over several steps before applying them (reference :13-57).

    fields: step, params (the U-Net's flat ParamStore), opt_state {count, mu (bf16), nu (fp32)}, grad_acc (flat), n_acc.
    """

    def __init__(self, unet, tx: AdamWConfig, process_group=None):
        self.unet = unet
        self.apply_fn = unet.forward
        self.params = unet.params
        self.tx = tx
        self.step = 0
        self.n_acc = 0
        self.grad_acc = unet.ensure_grads()
        n = self.params.flat.numel()
        dev = self.params.flat.device
        self.opt_state = {"count": 0, "mu": torch.zeros(n, dtype=torch.bfloat16, device=dev),
                          "nu": torch.zeros(n, dtype=torch.float32, device=dev)}
        self._sqnorm = torch.zeros(1, dtype=torch.float64, device=dev)
        self.process_group = process_group
        self.last_grad_norm = None

    @classmethod
    def create(cls, *, unet, tx, **kw):
        return cls(unet, tx, **kw)

    def world(self):
        if self.process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return torch.distributed.get_world_size(self.process_group)
        return 1

    def overlap_bucketer(self):
        """A GradBucketer for the backward pass that closes an optimizer update (a process group exists — world > 1, or the forced one-rank
        group of the RCCL tests — and DDPO_GRAD_OVERLAP != 0), else None: its
        bucketed all-reduce runs on a side stream behind the rest of that backward (training/distributed.GradBucketer); the blocking
        single all-reduce below stays the path of graph-replayed steps and of DDPO_GRAD_OVERLAP=0."""
        import os
        from .distributed import GradBucketer, _through_backend
        # world 1 only under DDPO_FORCE_DIST=1 (a one-rank RCCL group: the side stream, the per-bucket async all_reduce and finish() all
        # execute on hardware, tests/test_gpu_rccl_single_rank.py); a plain single process has no process group and keeps graph replay
        if not _through_backend() or os.environ.get("DDPO_GRAD_OVERLAP", "1") == "0":
            return None
        mib = float(os.environ.get("DDPO_GRAD_BUCKET_MIB", "256"))      # 256 MiB = 14 buckets over the 3.44 GB SD-1.5 gradient
        return GradBucketer(self.grad_acc.flat, bucket_numel=int(mib * (1 << 20)) // 4, group=self.process_group)

    def apply_gradients(self, *, grads=None, do_update, reduced=False):
        """`grads` were already accumulated into grad_acc by the backward kernels (grads is accepted for signature
        parity and must be None or grad_acc itself).  reduced=True: the sum over ranks is already in the buffer (GradBucketer)."""
        assert grads is None or grads is self.grad_acc
        if not do_update:
            self.n_acc += 1
            return self
        g = self.grad_acc.flat
        world = self.world()
        if world > 1 and not reduced:
            # lax.pmean(grad, "batch"): one all-reduce(sum) of the flat buffer, the mean is folded into inv_n below
            torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
        inv = 1.0 / ((self.n_acc + 1) * world)
        L.grad_sqnorm(g, self._sqnorm)
        t = self.opt_state["count"] + 1
        tx = self.tx
        L.adamw_bf16mu_step(self.params.flat, g, self.opt_state["mu"], self.opt_state["nu"], self._sqnorm, inv,
                            tx.learning_rate, tx.b1, tx.b2, tx.eps, tx.weight_decay, tx.max_grad_norm, t,
                            mu_decay_in_bf16=tx.mu_decay_in_bf16, zero_grad=True)
        self.last_grad_norm = torch.sqrt(self._sqnorm.clone()) * inv       # device scalar, no host sync
        if L.current_datapath() != "fp32":
            self.params.pack_bf16()                                        # refresh the bf16 hi/lo weight planes
        self.opt_state["count"] = t
        self.step += 1
        self.n_acc = 0
        return self


def _fwd_bwd(state, batch, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale, eta, clip_range, group=None, on_ready=None):
    """U-Net forward (cond + uncond as one batch), scoring-mode log-prob + PPO-clip forward/backward, U-Net backward
    (parameter gradients accumulate in place).  Pure device work: capturable into a HIP graph.
    `group`: rows per PPO micro-batch when the batch holds several micro-batches (train_steps_fused); info is then (k, 3)."""
    unet = state.unet
    lat = batch["latents"]
    b = lat.shape[0]
    ts = batch["ts"]
    tape = []
    if train_cfg:
        lat2 = torch.cat([lat, lat])
        ctx2 = torch.cat([batch["uncond_embeds"], batch["prompt_embeds"]]).contiguous()
        out = unet.forward(lat2, torch.cat([ts, ts]), ctx2, tape=tape)
        eps_u, eps_c = out[:b].contiguous(), out[b:].contiguous()
    else:
        out = unet.forward(lat, ts, batch["prompt_embeds"], tape=tape)
        eps_u, eps_c = None, out
    consts = noise_scheduler.kernel_consts(noise_scheduler_state, eta)
    d_c, d_u, per_sample, info = L.ddim_logprob_ppo_fwd_bwd(eps_c, eps_u, lat, batch["next_latents"], ts, batch["log_probs"],
                                                            batch["advantages"], guidance_scale, clip_range, train_cfg, consts,
                                                            group=group)
    d_out = torch.cat([d_u, d_c]) if train_cfg else d_c
    unet.backward(tape, d_out, on_ready=on_ready)
    return info, per_sample


_KEYS = ("latents", "next_latents", "ts", "log_probs", "advantages", "prompt_embeds", "uncond_embeds")


def _graphed_fwd_bwd(state, batch, sched_state, sched, train_cfg, guidance_scale, eta, clip_range, group=None):
    """Replay of _fwd_bwd as a captured HIP graph (one per batch geometry / hyper-parameter set): ~3000 kernel launches per
    micro-step become one graph launch.  Gradients still accumulate into the same flat buffer."""
    key = (tuple(batch["latents"].shape), tuple(batch["prompt_embeds"].shape), bool(train_cfg), float(guidance_scale), float(eta),
           float(clip_range), sched_state.num_inference_steps, L.current_datapath(), group)
    cache = state.__dict__.setdefault("_graphs", {})
    ent = cache.get(key)
    if ent == "eager":
        return _fwd_bwd(state, batch, sched_state, sched, train_cfg, guidance_scale, eta, clip_range, group)
    if ent is None:
        static = {k: batch[k].clone() for k in _KEYS}
        gflat = state.grad_acc.flat
        saved = gflat.clone()
        side = torch.cuda.Stream(gflat.device)
        side.wait_stream(torch.cuda.current_stream(gflat.device))
        with torch.cuda.stream(side):
            _fwd_bwd(state, static, sched_state, sched, train_cfg, guidance_scale, eta, clip_range, group)      # warm-up (allocations, attrs)
        torch.cuda.current_stream(gflat.device).wait_stream(side)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                info, per_sample = _fwd_bwd(state, static, sched_state, sched, train_cfg, guidance_scale, eta, clip_range, group)
        except Exception as exc:                # capture is an optimisation only: fall back to eager launches, loudly
            print(f"[ ddpo_amd ] WARNING: HIP-graph capture of train_step failed ({type(exc).__name__}: {exc}); launching eagerly")
            torch.cuda.synchronize(gflat.device)
            gflat.copy_(saved)
            cache[key] = "eager"
            return _fwd_bwd(state, batch, sched_state, sched, train_cfg, guidance_scale, eta, clip_range, group)
        gflat.copy_(saved)                      # warm-up passes must not leak into the accumulated gradients
        ent = (graph, static, info, per_sample)
        cache[key] = ent
    graph, static, info, per_sample = ent
    for k in _KEYS:
        static[k].copy_(batch[k])
    graph.replay()
    return info.clone(), per_sample.clone()


def train_step(state: AccumulatingTrainState, batch, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale,
               eta, clip_range, do_opt_update, jit=True):
    """One PPO micro-step (reference :63-146).  batch: latents, next_latents (b,4,h,w), ts (b,) int32, log_probs,
    advantages (b,), prompt_embeds, uncond_embeds (b,77,D) — device tensors.  Returns (state, info) with info a dict
    of device scalars {approx_kl, clipfrac, loss}."""
    assert isinstance(state, AccumulatingTrainState)
    b = batch["latents"].shape[0]
    assert b == batch["ts"].shape[0] == batch["next_latents"].shape[0] == batch["log_probs"].shape[0]
    dbatch = {k: batch[k].contiguous() for k in _KEYS if k in batch}
    dbatch["ts"] = dbatch["ts"].to(torch.int32)
    if not train_cfg and "uncond_embeds" not in dbatch:
        dbatch["uncond_embeds"] = dbatch["prompt_embeds"]
    bucketer = state.overlap_bucketer() if do_opt_update else None
    if bucketer is not None:
        info, per_sample = _fwd_bwd(state, dbatch, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale, eta, clip_range,
                                    on_ready=bucketer.ready)
        bucketer.finish()
    elif jit:
        info, per_sample = _graphed_fwd_bwd(state, dbatch, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale, eta, clip_range)
    else:
        info, per_sample = _fwd_bwd(state, dbatch, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale, eta, clip_range)
    state = state.apply_gradients(do_update=do_opt_update, reduced=bucketer is not None)
    return state, {"approx_kl": info[0], "clipfrac": info[1], "loss": info[2], "log_prob": per_sample[:, 0]}


def train_steps_fused(state: AccumulatingTrainState, batches, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale,
                      eta, clip_range, do_opt_update, jit=True):
    """k consecutive PPO micro-steps that see the SAME parameters — no optimizer update between them, i.e. any run of
    `train_step(..., do_opt_update=False)` calls optionally closed by one with `do_opt_update=True` — executed as ONE U-Net
    forward/backward over the concatenated rows.  The reference runs them one by one and sums their gradients in
    `AccumulatingTrainState.apply_gradients` (/root/reference/ddpo/training/policy_gradient.py:32-48; the entrypoint loop
    /root/reference/pipeline/policy_gradient.py:407-441 updates only at the last timestep of a mini-batch); summing inside
    one launch is the same arithmetic up to fp32 summation order, with 4x..16x more rows per GEMM (a micro-batch of 2
    samples x CFG is a U-Net batch of 4 — too small to fill 256 CUs).  Each micro-batch keeps its own mean loss
    (`ddpo_ddim_logprob_ppo_fwd_bwd_grouped`) and its own info row, and n_acc advances by k.

    batches: list of k dicts as for `train_step`, all with the same shapes.  `do_opt_update` applies after the LAST one.
    Returns (state, [info_0, ..., info_{k-1}])."""
    assert isinstance(state, AccumulatingTrainState)
    k = len(batches)
    assert k >= 1
    if k == 1:
        state, info = train_step(state, batches[0], noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale, eta,
                                 clip_range, do_opt_update, jit=jit)
        return state, [info]
    b = batches[0]["latents"].shape[0]
    for bt in batches:
        assert bt["latents"].shape == batches[0]["latents"].shape and bt["ts"].shape[0] == b
    if not train_cfg:
        batches = [dict(bt, uncond_embeds=bt.get("uncond_embeds", bt["prompt_embeds"])) for bt in batches]
    dbatch = {key: torch.cat([bt[key] for bt in batches]).contiguous() for key in _KEYS}
    dbatch["ts"] = dbatch["ts"].to(torch.int32)
    bucketer = state.overlap_bucketer() if do_opt_update else None
    if bucketer is not None:
        # the launch that closes an optimizer update runs eagerly, with the bucketed gradient all-reduce hanging on the backward's progress
        info, per_sample = _fwd_bwd(state, dbatch, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale, eta, clip_range, b,
                                    on_ready=bucketer.ready)
        bucketer.finish()
    else:
        fn = _graphed_fwd_bwd if jit else _fwd_bwd
        info, per_sample = fn(state, dbatch, noise_scheduler_state, noise_scheduler, train_cfg, guidance_scale, eta, clip_range, b)
    for _ in range(k - 1):
        state = state.apply_gradients(do_update=False)
    state = state.apply_gradients(do_update=do_opt_update, reduced=bucketer is not None)
    return state, [{"approx_kl": info[j, 0], "clipfrac": info[j, 1], "loss": info[j, 2], "log_prob": per_sample[j * b:(j + 1) * b, 0]}
                   for j in range(k)]
