#!/usr/bin/env python
"""Headline benchmark: sampled images/sec (512x512, 50 DDIM steps, CFG 5.0, eta 1.0, VAE decode included) on N MI355X.

A "step" = one pass of the sampling hot path over one per-GPU batch of `sample_batch_size` (8) prompts:
50 x [U-Net on 2B latents -> CFG -> Threefry noise -> DDIM step + log-prob] + VAE decode — BASELINE.json configs[1]
(compressed-animals geometry, SD-1.5 architecture, 512^2, 50 steps) with synthetic embeddings and random-init
weights (no checkpoints are reachable offline).  Weak scaling: every rank samples its own batch, no data-path
collective; value = all images of all ranks / max-over-ranks wall time.

Launch: `python bench.py --gpus N` starts its own N ranks (one per GPU, `torch.distributed.run`, backend nccl = RCCL) when it
is not already running under a launcher; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses
the launcher's ranks.  It refuses to run when the node has fewer GPUs than `--gpus`, or when WORLD_SIZE != --gpus.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     — dominant kernel (implicit-GEMM conv / GEMM on the MFMA datapath): algorithmic FLOPs / event-timed duration
  cpu_baseline — the CPU oracle (oracle/, torch-CPU) timed on a bounded sample of the same workload (rank 0, N=1)
  extra.train  — PPO train sample-timesteps/s of the same model (fused micro-steps, one AdamW update), rank-0 view
  rccl_ranks, allreduce — N > 1: world size seen by RCCL and a timed all-reduce of the flat fp32 gradient buffer

Modes: sample (headline) | train (PPO micro-steps) | epoch (one sample batch + its PPO micro-steps + optimizer updates with
the gradient all-reduce: what a DDPO epoch costs per GPU) | comm (the collective alone; also runs on CPU/gloo for the tests).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# algorithmic TFLOP per sample: U-Net forward at the benchmark latent size, VAE decode of one image
UNET_FWD_TFLOP = {("sd15", 512): 0.8033, ("sd21", 768): 2.1491}
VAE_TFLOP = {("sd15", 512): 2.5145, ("sd21", 768): 5.7543}
FP32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0                         # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak
N_UNET_PARAMS = {"sd15": 859_520_964, "sd21": 865_910_724}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="sd15", choices=["sd15", "sd21", "tiny", "tiny21"])
    ap.add_argument("--resolution", type=int, default=None, help="default: 512 (sd15), 768 (sd21), 64 (tiny)")
    ap.add_argument("--n-inference-steps", type=int, default=50)
    ap.add_argument("--sample-batch-size", type=int, default=8)
    ap.add_argument("--datapath", default=None, choices=["fp32", "bf16x3", "bf16", "f16mx"],
                    help="contraction datapath: exact-fp32 MFMA, bf16-split MFMA x3 (fp32-accurate to ~1e-5), single-pass bf16, f16mx (f16 MFMA + one "
                         "MX-scaled 8-bit MFMA for the cross terms on the long-reduction layers, bf16x3 elsewhere; 4e-5 on a U-Net forward).  "
                         "Default: DDPO_DATAPATH, else the datapath the entrypoints ship (ddpo_amd.lib.SHIPPED_DATAPATH = f16mx since round 4)")
    ap.add_argument("--mode", default="sample", choices=["sample", "train", "epoch", "comm"],
                    help="sample (headline): images/sec of the sampling hot path; train: PPO sample-timesteps/sec of train_step; "
                         "epoch: one sample batch + its PPO micro-steps + optimizer updates (gradient all-reduce included); "
                         "comm: only the gradient all-reduce")
    ap.add_argument("--train-batch-size", type=int, default=2)
    ap.add_argument("--train-fuse", type=int, default=None,
                    help="micro-steps per U-Net forward/backward (train_steps_fused); default: what the entrypoint uses for this geometry "
                         "(train_fuse_default: 16 at 64x64 latents, fewer for larger latents; DDPO_TRAIN_FUSE overrides); 1 = unfused")
    ap.add_argument("--backend", default=os.environ.get("DDPO_DIST_BACKEND"), help="nccl (= RCCL, default on GPUs) | gloo (CPU tests of --mode comm)")
    ap.add_argument("--comm-mib", type=float, default=None, help="--mode comm: buffer size in MiB (default: the flat fp32 gradient, 3.44 GB)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of an N-rank launch (needs --backend gloo): self-launch, rank environment, per-rank sampling keys, barrier / "
                         "max-over-ranks timing, the gradient all-reduce and the mode's JSON line with value = null and \"dry_run\": true — "
                         "NO engine work runs and nothing is measured (tests/test_bench_launch_cpu.py)")
    ap.add_argument("--no-graph", action="store_true", help="launch the U-Net kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train-extra", action="store_true", help="skip the short train-step measurement attached to the headline line")
    ap.add_argument("--no-alt-datapath-extra", "--no-f16mx-extra", dest="no_alt_extra", action="store_true",
                    help="skip the sampling measurement of the OTHER fp32-class datapath attached to the headline line (extra.bf16x3 / extra.f16mx)")
    args = ap.parse_args(argv)
    if args.datapath is None:
        from ddpo_amd import lib as _L
        args.datapath = _L.shipped_datapath()
    if args.resolution is None:
        args.resolution = {"sd15": 512, "sd21": 768}.get(args.model, 64)
    if args.train_fuse is None:
        from ddpo_amd.training.policy_gradient import train_fuse_default
        args.train_fuse = train_fuse_default(args.train_batch_size * 2, (args.resolution // 8) ** 2)      # train_cfg: 2 U-Net rows per sample
    return args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_self_launch(args, argv):
    """`python bench.py --gpus N` outside a launcher: check the node, then re-exec under torch.distributed.run with N ranks."""
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            raise SystemExit(f"bench.py: launched with WORLD_SIZE={os.environ['WORLD_SIZE']} but --gpus {args.gpus}; they must agree")
        return
    if args.gpus == 1:
        return
    on_cpu = (args.mode == "comm" or args.dry_run) and (args.backend == "gloo")
    if not on_cpu:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); "
                             f"refusing to run fewer ranks than asked (one process per GPU, RCCL)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    os.execvpe(cmd[0], cmd, env)


def cpu_baseline(args):
    """Times the oracle restatement (torch CPU, all host cores up to 32) on a bounded sample of the benchmark workload: ONE
    classifier-free guidance step of one image (the U-Net on a batch of 2 at the benchmark resolution, as the sampler runs it)
    and ONE VAE decode of one image; seconds/image = T x step + decode.  Protocol: 1 warm-up + 3 timed repeats
    of the U-Net step, median; the VAE decode (about a tenth of the U-Net share of an image) gets 1 warm-up + 1 timed run
    unless it is quick."""
    from oracle import unet as OU
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)      # torch-CPU stops scaling (and oversubscribes) beyond a few dozen threads
    torch.set_num_threads(threads)
    cfg = {"sd15": OU.SD15, "sd21": getattr(OU, "SD21", None), "tiny": OU.TINY, "tiny21": getattr(OU, "TINY21", OU.TINY)}[args.model]
    vcfg = OU.VAE_SD if args.model in ("sd15", "sd21") else OU.VAE_TINY
    g = torch.Generator().manual_seed(0)

    def synth(shapes):
        params = {}
        for name, shp in shapes.items():
            fan = int(np.prod(shp[:-1])) if name.endswith(".kernel") else 1
            params[name] = torch.randn(shp, generator=g) / (fan ** 0.5) if name.endswith(".kernel") else \
                (torch.ones(shp) if name.endswith(".scale") else torch.zeros(shp))
        return params
    params = synth(OU.unet_param_shapes(cfg))
    hw = args.resolution // 8
    x = torch.randn(1, 4, hw, hw, generator=g)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    t = torch.full((2,), 481, dtype=torch.int32)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        out = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            out.append(time.perf_counter() - t0)
        return out

    def unet_step():
        with torch.no_grad():
            OU.unet_forward(params, cfg, torch.cat([x, x]), t, ctx)          # [uncond; cond] in one batch, like the sampler
    unet_times = timed(unet_step, 1, 3)
    dt = statistics.median(unet_times)
    del params
    vparams = synth(OU.vae_decoder_param_shapes(vcfg))

    def vae_step():
        with torch.no_grad():
            OU.vae_decode(vparams, vcfg, x)
    t0 = time.perf_counter()
    vae_step()                                           # warm-up (also tells how long one decode takes)
    first = time.perf_counter() - t0
    vae_times = timed(vae_step, 0, 3 if first < 3.0 else 1)
    dt_vae = statistics.median(vae_times)
    T = args.n_inference_steps
    per_image = dt * T + dt_vae
    key = (args.model, args.resolution)
    gflops = (2 * UNET_FWD_TFLOP[key] + VAE_TFLOP[key]) * 1e3 / (dt + dt_vae) if key in UNET_FWD_TFLOP else None
    return {"value": 1.0 / per_image, "unit": "images/sec", "cores": threads, "host_cpu_count": ncpu, "threads": threads, "kind": "port",
            "sample": f"oracle (torch CPU, {threads} threads on a {ncpu}-core host): one CFG step of one image (U-Net on a batch of 2, {hw}x{hw} "
                      f"latents), 1 warm-up + 3 repeats, median {dt:.2f} s (runs {', '.join(f'{v:.2f}' for v in unet_times)}); one VAE decode to "
                      f"{args.resolution}x{args.resolution}, 1 warm-up + {len(vae_times)} run(s), median {dt_vae:.2f} s; "
                      f"seconds/image = {T} x step + decode",
            "seconds_per_cfg_step": dt, "seconds_per_vae_decode": dt_vae, "cpu_gflops": gflops}


class Comm:
    """World bookkeeping + the two collectives the benchmark itself needs (barrier, max over ranks)."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = None
        on_gpu = torch.cuda.is_available() and not ((args.mode == "comm" or args.dry_run) and args.backend == "gloo")
        self.dev = torch.device("cuda", self.local_rank) if on_gpu else torch.device("cpu")
        if on_gpu:
            torch.cuda.set_device(self.local_rank)
        # DDPO_FORCE_DIST=1: build the process group even for ONE rank — the only way to execute the RCCL path (communicator
        # bound to the device, all_reduce, barrier) on a 1-GPU box (tests/test_gpu_rccl_single_rank.py)
        if self.world > 1 or os.environ.get("DDPO_FORCE_DIST") == "1":
            import torch.distributed as dist
            self.dist = dist
            self.backend = args.backend or ("nccl" if on_gpu else "gloo")
            kw = {"device_id": self.dev} if self.backend == "nccl" else {}
            dist.init_process_group(self.backend, **kw)
            assert dist.get_world_size() == self.world == args.gpus

    def sync(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def time_allreduce(comm, numel, reps=3):
    """all_reduce(SUM) of an fp32 buffer of `numel` floats (the flat gradient): max-over-ranks milliseconds and bus bandwidth
    (2 (W-1)/W x bytes / time — what each rank's links carry in a ring / direct reduce-scatter + all-gather)."""
    if comm.dist is None:
        return None
    buf = torch.ones(int(numel), dtype=torch.float32, device=comm.dev)
    comm.dist.all_reduce(buf)                         # warm-up: communicator setup, first-touch
    comm.sync()
    times = []
    for _ in range(reps):
        buf.fill_(1.0)
        comm.sync()
        t0 = time.perf_counter()
        comm.dist.all_reduce(buf, op=comm.dist.ReduceOp.SUM)
        if comm.dev.type == "cuda":
            torch.cuda.synchronize()
        times.append(comm.max_over_ranks(time.perf_counter() - t0))
    ok = bool((buf == float(comm.world)).all().item())
    dt = statistics.median(times)
    nbytes = 4.0 * numel
    W = comm.world
    return {"bytes": nbytes, "ms": dt * 1e3, "algbw_GBps": nbytes / dt / 1e9, "busbw_GBps": 2.0 * (W - 1) / W * nbytes / dt / 1e9,
            "backend": comm.backend, "ranks": comm.dist.get_world_size(), "sum_correct": ok, "repeats": reps}


def bench_comm(args, comm):
    numel = int((args.comm_mib * (1 << 20)) // 4) if args.comm_mib else N_UNET_PARAMS.get(args.model, N_UNET_PARAMS["sd15"])
    if comm.dist is None:
        raise SystemExit("bench.py --mode comm needs --gpus >= 2")
    for _ in range(args.warmup):
        time_allreduce(comm, numel, reps=1)
    res = time_allreduce(comm, numel, reps=max(1, args.steps))
    if comm.rank == 0:
        print(json.dumps({"metric": "gradient all-reduce bus bandwidth", "value": res["busbw_GBps"], "unit": "GB/s", "n_gpus": comm.world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms"], "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"all_reduce(SUM) of {numel} fp32 (flat U-Net gradient buffer)", "parallelism": f"dp{comm.world}"},
                          "rccl_ranks": res["ranks"], "allreduce": res}), flush=True)
    comm.close()


def build_engine(args, comm):
    from ddpo_amd import lib as L
    from ddpo_amd.models.unet import UNet2DCondition, UNetConfig
    from ddpo_amd.models.vae import VAEDecoder, VAEConfig
    from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
    from ddpo_amd.diffusers_patch.pipeline_stable_diffusion import StableDiffusionPipeline
    L.load()
    L.DATAPATH = args.datapath
    dev = comm.dev
    ucfg = UNetConfig.named(args.model)
    unet = UNet2DCondition(ucfg, dev)
    unet.params.init_synthetic(seed=0)
    vae = VAEDecoder(VAEConfig.named("sd" if args.model in ("sd15", "sd21") else "tiny"), dev)
    vae.params.init_synthetic(seed=1)
    if L.DATAPATH != "fp32":
        unet.params.pack_bf16(bwd=args.mode in ("train", "epoch"))
        vae.params.pack_bf16(bwd=False)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1,
                          prediction_type=ucfg.prediction_type)
    state = sched.create_state(device=dev)
    pipe = StableDiffusionPipeline(unet, vae, sched)
    B = args.sample_batch_size
    g = torch.Generator().manual_seed(1 + comm.rank)
    emb = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g).to(dev)
    neg = torch.randn(1, 77, ucfg.cross_attention_dim, generator=torch.Generator().manual_seed(2)).expand(B, -1, -1).contiguous().to(dev)
    return L, ucfg, unet, vae, sched, state, pipe, emb, neg


def make_train_batch(args, comm, emb, neg, b):
    hw = args.resolution // 8
    g = torch.Generator().manual_seed(3 + comm.rank)
    lat = torch.randn(b, 4, hw, hw, generator=g).to(comm.dev)
    return {"latents": lat, "next_latents": (0.98 * lat + 0.05 * torch.randn(b, 4, hw, hw, generator=g).to(comm.dev)),
            "ts": torch.tensor([481, 21, 961, 241][:b], dtype=torch.int32, device=comm.dev),
            "log_probs": torch.full((b,), -1.0, device=comm.dev), "advantages": torch.tensor([0.7, -1.1, 0.3, -0.2][:b], device=comm.dev),
            "prompt_embeds": emb[:b].contiguous(), "uncond_embeds": neg[:b].contiguous()}


def measure_train(args, comm, L, unet, sched, state, emb, neg, steps, warmup):
    """`steps` timed launches of `train_fuse` fused micro-steps of `train_batch_size` sample-timesteps (train_cfg=True); every
    n_inference_steps micro-steps (one mini-batch of the entrypoint) close with the optimizer (gradient all-reduce over ranks +
    fused AdamW + weight re-pack)."""
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig, train_step, train_steps_fused
    if L.DATAPATH != "fp32":
        unet.params.pack_bf16(bwd=True)
    b = args.train_batch_size
    st = sched.set_timesteps(state, args.n_inference_steps)
    batch = make_train_batch(args, comm, emb, neg, b)
    tstate = AccumulatingTrainState(unet, AdamWConfig())
    fuse = max(1, args.train_fuse)
    k = 0
    T = int(args.n_inference_steps)      # the entrypoint closes a mini-batch (one optimizer update) after its T timesteps

    def one():
        nonlocal k
        k += fuse
        upd = k >= T
        if upd:
            k -= T
        if fuse == 1:
            train_step(tstate, batch, st, sched, True, 5.0, 1.0, 1e-4, do_opt_update=upd)
        else:
            train_steps_fused(tstate, [batch] * fuse, st, sched, True, 5.0, 1.0, 1e-4, do_opt_update=upd)
    for _ in range(warmup):
        one()
    comm.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    comm.sync()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    value = comm.world * b * fuse * steps / dt
    key = (args.model, args.resolution)
    tf = value * 6 * UNET_FWD_TFLOP[key] if key in UNET_FWD_TFLOP else None
    return {"metric": "PPO train sample-timesteps/sec (train_cfg)", "value": value, "unit": "sample-timesteps/sec", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "train_batch_size": b, "micro_steps_per_launch": fuse, "optimizer_update_every_micro_steps": T,
            "end_to_end_tflops": tf,
            "roofline": None if tf is None else {"bound": "mfma", "achieved": tf / comm.world, "peak": BF16_MFMA_PEAK_TFLOPS if args.datapath != "fp32" else FP32_MFMA_PEAK_TFLOPS,
                                                 "unit": "TFLOP/s", "frac": tf / comm.world / (BF16_MFMA_PEAK_TFLOPS if args.datapath != "fp32" else FP32_MFMA_PEAK_TFLOPS),
                                                 "note": "end-to-end: 6 x U-Net-forward algorithmic FLOPs per sample-timestep (2 fwd + 2 bwd) / wall time, per GPU"}}


# Measured context for `roofline` (round 6; never used as its `peak`): algorithmic TFLOP/s of a PURE MFMA stream per datapath — no loads, no LDS, random
# operands, every SIMD issuing back to back (tools/native/mfma_mix_bench, profiles/r06_mfma_mix_bench.log).  Under that load the chip settles at
# 1.58-1.68 GHz, not 2.4: single-pass bf16 sustains 1.60-1.64 PFLOP/s (97 % MFMA issue at the clock it holds), so the three-pass / f16 + MX-fp8
# operators top out at a third / about half of that however their operands are fed.
SUSTAINED_MFMA_ONLY_TFLOPS = {"bf16": 1615.0, "bf16x3": 539.0, "f16mx": 849.0}
SUSTAINED_MFMA_ONLY_NOTE = ("tools/native/mfma_mix_bench on MI355X, profiles/r06_mfma_mix_bench.log: pure MFMA streams (no memory traffic, random operands) hold "
                            "1.58-1.68 GHz; bf16 x1 1595-1636, f16 + MX-fp8 845-853, bf16x3 538-541 algorithmic TFLOP/s")
HBM_PEAK_GBPS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)


def measure_hbm_kernels(args, comm, L, ucfg, sched, state, unet):
    """Event-timed launches of the HBM-bound kernels of SURVEY.md §8(d) at the bench geometry, with §8(d)'s ALGORITHMIC bytes per unit:
    DDIM step 4 x 4*C*h*w B per sample-step (256 KiB at 64x64), PPO fwd+bwd 10 x 4*C*h*w (640 KiB), gradient norm 4 B/param,
    AdamW(bf16 mu) 24 B/param.  achieved = algorithmic bytes / average launch duration (HIP events on the launch stream)."""
    dev = comm.dev
    hw = args.resolution // 8
    chw = 4 * hw * hw
    st = sched.set_timesteps(state, args.n_inference_steps)
    consts = sched.kernel_consts(st, 1.0)
    out = {}

    def timed(fn, reps, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    def timed_graph(fn, reps):
        """`reps` launches captured into ONE HIP graph and replayed: what a launch costs on the DEVICE.  The microsecond kernels (DDIM step,
        PPO forward + backward) are launched from captured graphs / behind a 36 ms U-Net replay in the product; `timed` on them measures the
        Python + ctypes + launch rate of the host (9.5 us per call whatever the kernel does: a two-sample launch took as long as 64)."""
        fn(); torch.cuda.synchronize()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (3 * reps)

    def row(name, unit_bytes, units, dt, note):
        gbps = unit_bytes * units / dt / 1e9
        out[name] = {"bound": "hbm", "algorithmic_bytes_per_unit": unit_bytes, "units_per_launch": units, "avg_launch_us": dt * 1e6,
                     "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "note": note}

    g = torch.Generator().manual_seed(9)
    B = args.sample_batch_size
    mk = lambda n: torch.randn(n, 4, hw, hw, generator=g).to(dev)
    eu, ec, x, z = mk(B), mk(B), mk(B), mk(B)
    ts = torch.full((B,), 481, dtype=torch.int32, device=dev)
    xn, lp = torch.empty_like(x), torch.empty(B, device=dev)
    dt = timed_graph(lambda: L.ddim_step_fwd(eu, ec, x, z, ts, 5.0, consts, x_next=xn, logp=lp), 100)
    row("ddim_step", 4 * 4 * chw, B, dt, f"{B} samples per launch = {B * 16 * chw / 2**20:.1f} MiB: latency-bound by construction (one workgroup per "
        f"sample); the kernel also reads the pre-drawn Threefry noise (5 passes of {4 * chw // 1024} KiB in all); 100 launches replayed from one HIP graph")
    Bt = args.train_batch_size * max(1, args.train_fuse)
    eu, ec, x, x2 = mk(Bt), mk(Bt), mk(Bt), mk(Bt)
    ts = torch.full((Bt,), 481, dtype=torch.int32, device=dev)
    old_lp, adv = torch.full((Bt,), -1.0, device=dev), torch.randn(Bt, generator=g).to(dev)
    pre = L.ddim_logprob_ppo_fwd_bwd(ec, eu, x, x2, ts, old_lp, adv, 5.0, 1e-4, True, consts, group=args.train_batch_size)      # outputs allocated once, outside the timed calls
    dt = timed_graph(lambda: L.ddim_logprob_ppo_fwd_bwd(ec, eu, x, x2, ts, old_lp, adv, 5.0, 1e-4, True, consts, group=args.train_batch_size, out=pre), 100)
    row("ppo_fwd_bwd_grouped", 10 * 4 * chw, Bt, dt, f"{Bt} sample-timesteps per launch ({max(1, args.train_fuse)} fused micro-batches of {args.train_batch_size}), "
        "a cluster of 8 workgroups per sample-timestep (round 6: inputs read once, partial sums exchanged write-through, info row in the same launch); "
        "outputs pre-allocated; 100 launches replayed from one HIP graph (as train_steps_fused launches it)")
    n = unet.params.flat.numel()
    gbuf = torch.randn(n, generator=torch.Generator(device=dev).manual_seed(1), device=dev) * 1e-3
    sq = torch.zeros(1, dtype=torch.float64, device=dev)

    def norm():
        sq.zero_()
        L.grad_sqnorm(gbuf, sq)
    dt = timed(norm, 10)
    row("grad_sqnorm", 4, n, dt, "clip_by_global_norm's reduction over the flat gradient (units = parameters)")
    pbuf = unet.params.flat.clone()
    mu = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    nu = torch.zeros(n, dtype=torch.float32, device=dev)
    step = [0]

    def adam():
        step[0] += 1
        L.adamw_bf16mu_step(pbuf, gbuf, mu, nu, sq, 1.0, 1e-5, 0.9, 0.999, 1e-8, 1e-4, 1.0, step[0], zero_grad=False)
    dt = timed(adam, 10)
    row("adamw_bf16mu", 24, n, dt, "clip + AdamW(bf16 mu) + weight decay over the flat buffers on a private copy of the parameters (units = parameters; "
        "zero_grad off so that the gradient stays defined over the repeats: the shipped call also writes the 4 B/param of zeros)")
    return out


def roofline_traffic(dom):
    """PMC-measured HBM bytes per launch of the dominant kernel family from profiles/roofline_traffic.json — rocprofv3 cannot run
    inside this process, so the figure is collected by tools/pmc_unet_traffic.sh over exactly the launches the roofline pass
    event-times and STAMPED with the git commit / date of that run.  It is refused (null + reason) when the stamp is missing or any
    kernel source is newer than the collection (the k-loop may have changed since)."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if not os.path.exists(tpath):
        return None, "profiles/roofline_traffic.json not found"
    tall = json.load(open(tpath))
    tj = tall.get(dom + "_unet") or tall.get(dom + "_r02_unet") or tall.get(dom)
    if not tj:
        return None, f"no entry for datapath {dom}"
    stamp = tj.get("collected_unix")
    if stamp is None:
        return None, "entry carries no collection stamp (collected before round 3): re-run tools/pmc_unet_traffic.sh"
    srcs = [os.path.join(ROOT, "ddpo_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "ddpo_amd", "csrc")) if f.endswith((".hip", ".h"))]
    hashes = tj.get("csrc_sha256")
    if hashes:
        import hashlib
        cur = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16] for f in srcs}
        changed = sorted(k for k in ("gemm_bf16.hip", "common.h") if cur.get(k) != hashes.get(k))
        if changed:
            return None, f"kernel sources changed since the PMC run of {tj.get('collected_date')} (commit {tj.get('git_commit')}): {', '.join(changed)}"
    note = tj.get("note", "") + f" [collected {tj.get('collected_date')}, commit {tj.get('git_commit')}, {tj.get('launches')} launches]"
    return tj["traffic_bytes_per_launch"], note


def bench_train(args, comm):
    L, ucfg, unet, vae, sched, state, pipe, emb, neg = build_engine(args, comm)
    res = measure_train(args, comm, L, unet, sched, state, emb, neg, args.steps, args.warmup)
    ar = time_allreduce(comm, unet.params.flat.numel()) if comm.dist is not None else None
    if comm.rank == 0:
        print(json.dumps(train_line(args, comm, res, ar)), flush=True)
    comm.close()


def bench_epoch(args, comm):
    """One DDPO epoch per GPU at the reference defaults' shape: 1 sample batch of 8 images (50 steps, CFG, VAE decode), then
    for each of the 8/2 = 4 mini-batches its 50 PPO micro-steps (fused 10 at a time) closed by ONE optimizer update (gradient
    all-reduce over ranks + AdamW): 200 micro-steps, 4 updates.  Reward evaluation (host JPEG) is not part of the timed region."""
    from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig, train_steps_fused
    from ddpo_amd.utils import prng
    L, ucfg, unet, vae, sched, state, pipe, emb, neg = build_engine(args, comm)
    T, B, b = args.n_inference_steps, args.sample_batch_size, args.train_batch_size
    fuse = max(1, min(args.train_fuse, T))
    st = sched.set_timesteps(state, T)
    tstate = AccumulatingTrainState(unet, AdamWConfig())
    _, sample_rng = prng.split(prng.PRNGKey(0))
    adv = torch.randn(B, generator=torch.Generator().manual_seed(5 + comm.rank)).clamp_(-10, 10).to(comm.dev)

    def one_epoch():
        nonlocal sample_rng
        sample_rng, sample_seed = prng.split(sample_rng)
        key = prng.split(sample_seed, comm.world)[comm.rank]
        final, lat, nxt, lps, ts = pipe(emb, neg, {"unet": unet.params, "scheduler": state}, key, T, height=args.resolution,
                                        width=args.resolution, guidance_scale=5.0, eta=1.0, jit=not args.no_graph)
        img = vae.decode(final)
        for i in range(B // b):
            sl = slice(i * b, (i + 1) * b)
            for j0 in range(0, T, fuse):
                js = range(j0, min(j0 + fuse, T))
                batches = [{"prompt_embeds": emb[sl], "uncond_embeds": neg[sl], "advantages": adv[sl], "latents": lat[sl, j].contiguous(),
                            "next_latents": nxt[sl, j].contiguous(), "log_probs": lps[sl, j].contiguous(), "ts": ts[sl, j].contiguous()} for j in js]
                train_steps_fused(tstate, batches, st, sched, True, 5.0, 1.0, 1e-4, do_opt_update=(js[-1] == T - 1))
        return img
    for _ in range(args.warmup):
        one_epoch()
    comm.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = one_epoch()
    comm.sync()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(img).all()
    ar = time_allreduce(comm, unet.params.flat.numel()) if comm.dist is not None else None
    if comm.rank == 0:
        print(json.dumps(epoch_line(args, comm, dt, fuse, ar)), flush=True)
    comm.close()


def epoch_line(args, comm, dt, fuse, ar):
    """The `--mode epoch` JSON line (dt = max-over-ranks seconds of the timed epochs; None in a dry run)."""
    T, B, b = args.n_inference_steps, args.sample_batch_size, args.train_batch_size
    n_upd = B // b
    per = (lambda x: None) if dt is None else (lambda x: x / dt)
    return {"metric": "DDPO epochs/sec per job (8 images sampled + 200 PPO micro-steps + 4 AdamW updates per GPU)",
            "value": per(args.steps), "unit": "epochs/sec", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None if dt is None else dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.datapath, "data": "synthetic",
            "config": {"workload": f"{args.model} {args.resolution}^2: {B} images x {T} DDIM steps + VAE decode, then {B // b} mini-batches x {T} "
                                   f"PPO micro-steps (train_cfg, {fuse} fused per launch), {n_upd} optimizer updates each with one "
                                   f"all-reduce of the flat gradient", "parallelism": f"dp{comm.world}", "global_batch": comm.world * B},
            "images_per_sec": per(comm.world * B * args.steps), "sample_timesteps_per_sec": per(comm.world * B * T * args.steps),
            "rccl_ranks": comm.world if comm.dist is not None else None, "allreduce": ar}


def train_line(args, comm, res, ar):
    """The `--mode train` JSON line."""
    return {"metric": f"PPO train sample-timesteps/sec (train_cfg, {args.resolution}^2)", "value": res["value"], "unit": "sample-timesteps/sec",
            "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.datapath, "data": "synthetic",
            "config": {"workload": f"train_step, {args.model}, train_batch_size {args.train_batch_size}/GPU, train_cfg, "
                                   f"{res['micro_steps_per_launch']} micro-step(s) per U-Net forward/backward, optimizer update every {res['optimizer_update_every_micro_steps']} micro-steps",
                       "parallelism": f"dp{comm.world}"},
            "end_to_end_tflops": res["end_to_end_tflops"], "roofline": res["roofline"],
            "rccl_ranks": comm.world if comm.dist is not None else None, "allreduce": ar}


def dry_run(args, comm):
    """`--dry-run --backend gloo`: everything of an N-rank bench EXCEPT the engine — so that the first launch on an 8-GPU node does not
    spend its run on a rank-environment or JSON bug.  Checks: one key per rank from the reference key tree (pairwise distinct over ranks),
    barrier + max-over-ranks, the all-reduce of a gradient-shaped buffer; rank 0 prints the mode's line with value = null."""
    from ddpo_amd.utils import prng
    from ddpo_amd.training import distributed as D
    if comm.dev.type != "cpu":
        raise SystemExit("bench.py --dry-run is the CPU rehearsal: pass --backend gloo")
    _, sample_rng = prng.split(prng.PRNGKey(0))
    sample_rng, sample_seed = prng.split(sample_rng)
    key = np.asarray(prng.split(sample_seed, comm.world)[comm.rank], dtype=np.uint32)
    keys = torch.from_numpy(key.astype(np.int64)).reshape(1, -1)
    if comm.dist is not None:
        allk = [torch.zeros_like(keys) for _ in range(comm.world)]
        comm.dist.all_gather(allk, keys)
        keys = torch.cat(allk)
    assert len({tuple(r.tolist()) for r in keys}) == comm.world, "ranks must sample from pairwise distinct keys"
    comm.sync()
    t0 = time.perf_counter()
    comm.sync()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    assert dt >= 0.0
    numel = int((args.comm_mib or 1.0) * (1 << 20) // 4)
    ar = time_allreduce(comm, numel, reps=1)
    if comm.rank == 0:
        T, b = args.n_inference_steps, args.train_batch_size
        if args.mode == "epoch":
            line = epoch_line(args, comm, None, max(1, min(args.train_fuse, T)), ar)
        elif args.mode == "train":
            fuse = max(1, min(args.train_fuse, T))
            line = train_line(args, comm, {"value": None, "ms_per_step": None, "micro_steps_per_launch": fuse,
                                           "optimizer_update_every_micro_steps": T, "end_to_end_tflops": None, "roofline": None}, ar)
        else:
            line = sample_line(args, comm, None, None, None, ar, {}, None)
        line["dry_run"] = True
        line["data"] = "none (dry run: no engine work, nothing measured)"
        print(json.dumps(line), flush=True)
    comm.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    maybe_self_launch(args, argv)
    if args.mode != "comm" and not args.dry_run and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the DDPO engine has no CPU path)")
    comm = Comm(args)
    if args.mode == "comm":
        return bench_comm(args, comm)
    if args.dry_run:
        return dry_run(args, comm)
    if args.mode == "train":
        return bench_train(args, comm)
    if args.mode == "epoch":
        return bench_epoch(args, comm)

    from ddpo_amd.utils import prng
    L, ucfg, unet, vae, sched, state, pipe, emb, neg = build_engine(args, comm)
    world, rank, dev = comm.world, comm.rank, comm.dev
    B = args.sample_batch_size
    # reference key tree (pipeline/policy_gradient.py:51,201,244-245): rank r uses row r of split(sample_seed, n_devices)
    rng = prng.PRNGKey(0)
    _, sample_rng = prng.split(rng)

    def one_step():
        nonlocal sample_rng
        sample_rng, sample_seed = prng.split(sample_rng)
        key = prng.split(sample_seed, world)[rank]
        final, lat, nxt, lps, ts = pipe(emb, neg, {"unet": unet.params, "scheduler": state}, key, args.n_inference_steps,
                                        height=args.resolution, width=args.resolution, guidance_scale=5.0, eta=1.0,
                                        jit=not args.no_graph)
        img = vae.decode(final)
        return img, lps

    for _ in range(args.warmup):
        one_step()
    comm.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img, lps = one_step()
    comm.sync()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(img).all() and torch.isfinite(lps).all()
    images = world * B * args.steps
    value = images / dt

    roofline = None
    if not args.no_roofline:
        # event-timed pass over the dominant kernel family (every ddpo_gemm_conv_fwd launch of one U-Net forward
        # on 2B latents); events are recorded on the stream the kernels are launched on.
        lat2 = torch.randn(2 * B, 4, args.resolution // 8, args.resolution // 8, device=dev)
        ts2 = torch.full((2 * B,), 481, dtype=torch.int32, device=dev)
        ctx2 = torch.cat([neg, emb])
        unet(lat2, ts2, ctx2)
        torch.cuda.synchronize()
        L.PROFILE = []
        unet(lat2, ts2, ctx2)
        torch.cuda.synchronize()
        recs, L.PROFILE = L.PROFILE, None
        # the dominant kernel family = every launch of the bf16-MFMA GEMM / conv template of this datapath (under f16mx: its f16 + MX-fp8
        # instantiation on the long reductions AND the bf16x3 instantiations on the rest); the exact-fp32 launches (conv_in / conv_out) are not in it
        fams = {"fp32": ("fp32",), "bf16": ("bf16",), "bf16x3": ("bf16x3",), "f16mx": ("f16mx", "bf16x3")}[args.datapath]
        dom = args.datapath if any(r[3] in fams for r in recs) else "fp32"
        recs = [r for r in recs if r[3] in (fams if dom != "fp32" else ("fp32",))]
        flops = sum(r[2] for r in recs)
        ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        achieved = flops / (ms * 1e-3) / 1e12
        # matrix-pipe pass-equivalents per algorithmic FLOP, FLOP-weighted over the family: 3 on bf16x3 launches, 2 on f16mx launches (2 f16 MFMAs
        # + 1 double-rate 8-bit MFMA per block and k-tile), 1 on single-pass kernels
        pw = {"fp32": 1, "bf16": 1, "bf16x3": 3, "f16mx": 2}
        passes = sum(pw[r[3]] * r[2] for r in recs) / max(flops, 1.0)
        by_fam = {}
        for r in recs:
            e = by_fam.setdefault(r[3], [0, 0.0, 0.0])
            e[0] += 1; e[1] += r[2]; e[2] += r[0].elapsed_time(r[1])
        peak = FP32_MFMA_PEAK_TFLOPS if dom == "fp32" else BF16_MFMA_PEAK_TFLOPS
        kname = {"fp32": "gemm_conv_kernel (v_mfma_f32_32x32x2_f32)",
                 "f16mx": "gemm_conv_bf16_buf_kernel<256x320 | 128x320 | 128x128 | 128x64>: NPASS=4 (v_mfma_f32_32x32x16_f16 + v_mfma_scale_f32_32x32x64_f8f6f4) on K >= 2560 "
                          "layers, NPASS=3 (v_mfma_f32_32x32x16_bf16) elsewhere"}.get(
            dom, f"gemm_conv_bf16_buf_kernel<256x320 | 128x320 | 128x128 | 128x64, NPASS={pw[dom]}> (v_mfma_f32_32x32x16_bf16)")
        traffic, traffic_note = roofline_traffic(dom)
        roofline = {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                    "frac": achieved / peak, "traffic": traffic, "traffic_note": traffic_note,
                    "algorithmic_bytes_per_launch": sum(r[4] for r in recs) / max(len(recs), 1),
                    "launches": len(recs), "avg_launch_ms": ms / max(len(recs), 1),
                    "algorithmic_gflop_per_launch": flops / max(len(recs), 1) / 1e9,
                    "mfma_passes_per_algorithmic_flop": passes, "mfma_issue_frac": passes * achieved / peak,
                    "by_instantiation": {k: {"launches": v[0], "tflops": v[1] / (v[2] * 1e-3) / 1e12 if v[2] else None, "ms": v[2],
                                             "frac_of_sustained_mfma_only": (v[1] / (v[2] * 1e-3) / 1e12 / SUSTAINED_MFMA_ONLY_TFLOPS[k])
                                             if (v[2] and k in SUSTAINED_MFMA_ONLY_TFLOPS) else None} for k, v in by_fam.items()},
                    # context, NOT the roofline: what a pure stream of this datapath's MFMAs (no memory traffic, random operands) sustains on this chip
                    "sustained_mfma_only": {"tflops": SUSTAINED_MFMA_ONLY_TFLOPS, "source": SUSTAINED_MFMA_ONLY_NOTE}}
    ar = time_allreduce(comm, unet.params.flat.numel()) if comm.dist is not None else None
    extra = {}
    if not args.no_train_extra and args.model in ("sd15", "sd21"):
        # the PPO half of an epoch (the wall-clock bottleneck of "reward vs wall-clock"), reported next to the headline number
        try:
            extra["train"] = measure_train(args, comm, L, unet, sched, state, emb, neg, steps=12, warmup=2)
        except Exception as exc:          # the headline line must survive a failure of the secondary measurement
            extra["train"] = {"error": f"{type(exc).__name__}: {exc}"}
    if not args.no_roofline:
        try:
            extra["hbm_kernels"] = measure_hbm_kernels(args, comm, L, ucfg, sched, state, unet)
        except Exception as exc:
            extra["hbm_kernels"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank != 0:
        comm.close()
        return
    alt = {"f16mx": "bf16x3", "bf16x3": "f16mx"}.get(args.datapath)
    if world == 1 and alt and not args.no_alt_extra and not args.no_train_extra and args.model == "sd15":
        # the OTHER fp32-class datapath on the same workload, in a fresh process: reported NEXT to the headline, never as it
        # (f16mx ships since round 4; bf16x3 is the selectable three-pass datapath)
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--datapath", alt, "--steps", str(args.steps), "--warmup", str(args.warmup),
                   "--no-cpu-baseline", "--no-train-extra", "--no-roofline", "--sample-batch-size", str(args.sample_batch_size)]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
            pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in pr.stdout.splitlines() if l.startswith('{"metric"')][-1]
            dj = json.loads(line)
            extra[alt] = {"value": dj["value"], "unit": dj["unit"], "ms_per_step": dj["ms_per_step"], "dtype": dj["dtype"],
                          "note": f"the same workload with --datapath {alt} (U-Net forward error against float64: f16mx 4.2e-5, bf16x3 2.0e-5)"}
        except Exception as exc:
            extra[alt] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and args.datapath == "f16mx" and L.MX_CROSS and not args.no_alt_extra and not args.no_train_extra and args.model == "sd15":
        # opt-in, NOT the headline and not what ships: the f16mx layers without their cross terms (one f16 pass per product: the reference's own
        # arithmetic class — its TPUs run one bf16 pass — at 11 significant bits).  What the cross terms' matrix work and bytes cost on this
        # power-bound chip; its parity margins (outside the 1e-3 contract on the gradients) are in profiles/r06_parity_f16x1.log.
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--datapath", "f16mx", "--steps", str(args.steps), "--warmup", str(args.warmup),
                   "--no-cpu-baseline", "--no-train-extra", "--no-roofline", "--sample-batch-size", str(args.sample_batch_size)]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
            env["DDPO_MX_CROSS"] = "0"
            pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            dj = json.loads([l for l in pr.stdout.splitlines() if l.startswith('{"metric"')][-1])
            extra["f16x1"] = {"value": dj["value"], "unit": dj["unit"], "ms_per_step": dj["ms_per_step"],
                              "dtype": "f32 storage; K >= 2560 layers: ONE f16 MFMA pass per product (no cross terms); elsewhere as the headline",
                              "note": "DDPO_MX_CROSS=0: opt-in, NOT shipped, not the headline — it is OUTSIDE the north-star contract: U-Net forward 7.3e-4 rms against "
                                      "float64 (shipped operator 4e-5), latents 7e-4 along the trajectory (6.6e-5), and at full size the train step's block "
                                      "gradient norms 2.4e-3 and gradient vector 2.7e-2 against the 1e-3 / 2e-3 gates the shipped operator holds at 1.5e-4 / "
                                      "1.2e-3 (profiles/r06_parity_f16x1.log).  Reported to price the cross terms: what they cost in throughput on this power-bound chip"}
        except Exception as exc:
            extra["f16x1"] = {"error": f"{type(exc).__name__}: {exc}"}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    print(json.dumps(sample_line(args, comm, value, dt, roofline, ar, extra, cpu)), flush=True)
    comm.close()


def sample_line(args, comm, value, dt, roofline, ar, extra, cpu):
    """The headline JSON line (value = whole-job images/sec, dt = max-over-ranks seconds of the timed steps; both None in a dry run)."""
    world, B = comm.world, args.sample_batch_size
    key = (args.model, args.resolution)
    tflop_per_image = args.n_inference_steps * 2 * UNET_FWD_TFLOP[key] + VAE_TFLOP[key] if key in UNET_FWD_TFLOP else None
    out = {
        "metric": f"sampled images/sec ({args.resolution}^2, {args.n_inference_steps} DDIM steps)", "value": value, "unit": "images/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None if dt is None else dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "f32 (bf16x3-split MFMA)", "bf16": "bf16 products, f32 accumulate",
                                                                 "f16mx": "f32 (f16mx: f16 MFMA + MX-fp8 cross terms on the long-reduction conv/GEMM layers, bf16x3-split MFMA elsewhere)"
                                                                          if os.environ.get("DDPO_MX_CROSS", "1") == "1" else
                                                                          "f32 (DDPO_MX_CROSS=0, opt-in: ONE f16 MFMA pass on the long-reduction conv/GEMM layers, bf16x3-split MFMA elsewhere)"}[args.datapath],
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{1 if args.model == 'sd15' else 4}]: {'compressed-animals' if args.model == 'sd15' else 'neg_jpeg'} geometry, "
                               f"{args.model} U-Net+VAE (random init), "
                               f"{args.resolution}x{args.resolution}, {args.n_inference_steps} DDIM steps, CFG 5.0, eta 1.0, "
                               f"sample_batch_size {B}/GPU, VAE decode included",
                   "datapath": {"fp32": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)",
                                "bf16x3": "conv/GEMM: bf16x3-split MFMA (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulate, ~1e-5 rel; "
                                          "= XLA HIGH, the reference ran TPU DEFAULT = 1 pass); attention (d in 40/64/80): same split; d=160 attention and norms: exact fp32",
                                "bf16": "conv/GEMM: single-pass bf16 MFMA, fp32 accumulate (= XLA TPU DEFAULT precision)",
                                "f16mx": "shipped default (lib.SHIPPED_DATAPATH): conv/GEMM layers with a reduction K >= 2560 run a_h*b_h on the f16 MFMA + ONE MX-scaled "
                                         "8-bit MFMA carrying both cross terms (a_h8*b_l8 + a_l8*b_h8); attention (d in 8/16/40/64/80) on the f16p operators (scores bf16x3, "
                                         "probabilities one f16 term against V f16 hi/lo: 2 second-product passes; backward likewise); short reductions and ALL data / weight "
                                         "gradients as under bf16x3; ~4e-5 rel on an SD-1.5 U-Net forward against float64 (bf16x3 2e-5, north-star gate 1e-3)"}[args.datapath],
                   "parallelism": f"dp{world}",
                   "global_batch": world * B},
        "end_to_end_tflops": None if (tflop_per_image is None or value is None) else value * tflop_per_image,
        "roofline": roofline, "cpu_baseline": cpu,
        "rccl_ranks": world if comm.dist is not None else None, "allreduce": ar, "extra": extra,
    }
    return out


if __name__ == "__main__":
    main()
