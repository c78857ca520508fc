#!/usr/bin/env python
"""DDPO outer loop on MI355X — drop-in for the reference entrypoint.

    python pipeline/policy_gradient.py --dataset compressed-animals [--key value ...]
    torchrun --nproc-per-node 8 pipeline/policy_gradient.py --dataset a-animals        (one process per GPU)

Per epoch: sample DDIM trajectories (HIP sampler) -> decode -> reward callbacks (async thread) -> all-gather rewards
-> advantages (per-prompt tracker or global normalisation) -> shuffle over batch and, per sample, over time ->
PPO-clip updates on the stored log-probs (HIP forward/backward, one RCCL all-reduce + fused AdamW per update)
-> .npy run artefacts, checkpoints, reward curve.

Mirrors /root/reference/pipeline/policy_gradient.py:45-480 step for step (same flags, same artefacts under
logs/<savepath>/, same Python/numpy RNG call order for prompts and shuffles, same JAX key tree for the noise).
Data parallelism follows the reference's multi-host mode with ONE local device per process: seed + rank
(ddpo/utils/parser.py:177), per-process prompts, rewards all-gathered, this rank's slice of the advantages;
DDPO_DP_SEMANTICS=single_host reproduces the single-host N-device run instead (ddpo_amd/training/dp.py).
Differences, all deliberate: trajectories stay in HBM instead of round-tripping through host numpy (:292-295,:415-423);
gradients are all-reduced once per optimizer update instead of every micro-step (ddpo/training/policy_gradient.py:141);
`info` is fetched once per inner epoch instead of a blocking device_get + assert_equal per step (:442-445).
"""
import json
import os
import random
import sys
import time
from concurrent import futures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from ddpo_amd import training, utils
from ddpo_amd.training import distributed as D
from ddpo_amd.training.dp import DataParallel
from ddpo_amd.training.policy_gradient import (AccumulatingTrainState, AdamWConfig, train_fuse_default, train_step,
                                               train_steps_fused)
from ddpo_amd.utils import prng
from ddpo_amd.utils.serialization import load_params_file, load_resume, load_unet, save_checkpoint, save_rank_resume
from ddpo_amd.utils.stat_tracking import PerPromptStatTracker
from ddpo_amd.models.text import make_uncond_text


def _flag(value, default):
    """CLI flags arrive as strings ("False", "0"); None = not given."""
    if value is None:
        return default
    return value if isinstance(value, bool) else str(value).lower() not in ("0", "false", "no")


class Parser(utils.Parser):
    config: str = "config.base"
    dataset: str = "consistent_imagenet"


def main(argv=None):
    worker_id, n_workers = D.init()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("pipeline/policy_gradient.py needs a GPU: the DDPO engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # contraction datapath of the HIP kernels: lib.SHIPPED_DATAPATH (f16mx since round 4); DDPO_DATAPATH=bf16x3 / fp32 select the others
    from ddpo_amd import lib as L
    L.DATAPATH = L.shipped_datapath()

    # which reference run the ranks reproduce: one PROCESS each (default) or one DEVICE each of a single-host run
    # (DDPO_DP_SEMANTICS=single_host: identical seeds, global prompt / permutation streams, trajectories all-gathered; training/dp.py)
    dp = DataParallel(rank=worker_id, world=n_workers)
    args = Parser(argv).parse_args("pg", process_index=dp.seed_process_index)
    utils.init_logging("policy_gradient", args.verbose)

    rng = prng.PRNGKey(args.seed)
    n_devices = 1                                            # one process drives one GPU
    train_worker_batch_size = n_devices * args.train_batch_size
    train_pod_batch_size = train_worker_batch_size * n_workers
    train_effective_batch_size = train_pod_batch_size * args.train_accumulation_steps
    sample_worker_batch_size = n_devices * args.sample_batch_size
    sample_pod_batch_size = sample_worker_batch_size * n_workers
    total_samples_per_epoch = args.num_sample_batches_per_epoch * sample_pod_batch_size
    print(f"[ policy_gradient ] local devices: {n_devices} | number of workers: {n_workers}")
    print(f"[ policy_gradient ] sample worker batch size: {sample_worker_batch_size} | sample pod batch size: {sample_pod_batch_size}")
    print(f"[ policy_gradient ] train worker batch size: {train_worker_batch_size} | train pod batch size: {train_pod_batch_size} | "
          f"train accumulated batch size: {train_effective_batch_size}")
    print(f"[ policy_gradient ] number of sample batches per epoch: {args.num_sample_batches_per_epoch}")
    print(f"[ policy_gradient ] total number of samples per epoch: {total_samples_per_epoch}")
    print(f"[ policy_gradient ] number of gradient updates per inner epoch: {total_samples_per_epoch // train_effective_batch_size}")
    assert args.sample_batch_size >= args.train_batch_size
    assert args.sample_batch_size % args.train_batch_size == 0
    assert total_samples_per_epoch % train_effective_batch_size == 0

    localpath = "logs/" + args.savepath.replace("gs://", "")
    os.makedirs(localpath, exist_ok=True)

    # --------------------------------- models ---------------------------------#
    print("loading models...")
    pipeline, params = load_unet(None, epoch=args.load_epoch, pretrained_model=args.pretrained_model, dtype=args.dtype,
                                 cache=args.cache, device=dev, seed=0)
    with open(f"{localpath}/args.json", "w") as f:      # after loading: says which weights the run really started from
        json.dump(dict(args._dict, synthetic_weights=bool(pipeline.synthetic_weights), weights_source=pipeline.weights_source,
                       param_dtype=pipeline.param_dtype, datapath=L.DATAPATH), f, indent=4, default=str)
    pipeline.safety_checker = None
    unet, vae = pipeline.unet, pipeline.vae
    noise_scheduler_state = pipeline.scheduler.set_timesteps(params["scheduler"], num_inference_steps=args.n_inference_steps)

    # ------------------------------- optimizer --------------------------------#
    print("initializing train state...")
    if args.optimizer != "adamw":
        raise NotImplementedError("only the adamw optimizer of the reference configs is implemented")
    state = AccumulatingTrainState(unet, AdamWConfig(learning_rate=args.learning_rate, b1=args.beta1, b2=args.beta2,
                                                     eps=args.epsilon, weight_decay=args.weight_decay,
                                                     max_grad_norm=args.max_grad_norm,
                                                     # optax.adamw(mu_dtype=bfloat16) forms `b1 * mu` with mu in bf16: JAX's weak-type
                                                     # promotion keeps the ARRAY's dtype, so the product (and b1 itself, 0.9 -> 0.8984375) is
                                                     # rounded to bf16 before the f32 `(1 - b1) * g` is added — the default here and in
                                                     # oracle/optim.py.  `--mu_decay_in_bf16 False` / DDPO_MU_DECAY_IN_BF16=0 selects the all-f32
                                                     # reading (decay in f32, one rounding at the store) should a real optax run disagree.
                                                     mu_decay_in_bf16=_flag(getattr(args, "mu_decay_in_bf16", None),
                                                                            os.environ.get("DDPO_MU_DECAY_IN_BF16", "1") != "0")))
    sampling_scheduler_params = params["scheduler"]

    timer = utils.Timer()
    text_encode = params["text_encoder"]
    uncond_prompt_ids = make_uncond_text(pipeline.tokenizer, 1)
    timer()
    uncond_prompt_embeds = text_encode(uncond_prompt_ids)[0]
    print(f"[ embed uncond prompts ] in {timer():.2f}s")
    sample_uncond_prompt_embeds = uncond_prompt_embeds.unsqueeze(0).expand(args.sample_batch_size, -1, -1).contiguous()
    train_uncond_prompt_embeds = sample_uncond_prompt_embeds[: args.train_batch_size].contiguous()

    train_rng, sample_rng = prng.split(rng)

    # ------------------------------ callbacks -------------------------------#
    callback_fns = {args.filter_field: training.callback_fns[args.filter_field]()}
    executor = futures.ThreadPoolExecutor(max_workers=2)     # rewards run next to the sampling of the following batch

    per_prompt_stats = None
    if args.per_prompt_stats_bufsize is not None:
        per_prompt_stats = PerPromptStatTracker(args.per_prompt_stats_bufsize, args.per_prompt_stats_min_count)

    mean_rewards, std_rewards, wall = [], [], []
    t_start = time.time()
    start_epoch = 0
    if os.environ.get("DDPO_RESUME"):
        # not in the reference (it never loads a policy-gradient run): continue from <run>/checkpoints — parameters, AdamW moments
        # and count, the sampling key, the host RNG streams of this rank, the per-prompt tracker and the reward history
        rs = load_resume(os.environ["DDPO_RESUME"], worker_id, os.environ.get("DDPO_RESUME_EPOCH"))
        load_params_file(state.params, rs["params_path"])
        if L.DATAPATH != "fp32":
            state.params.pack_bf16()
        state.opt_state["mu"].copy_(rs["mu"])
        state.opt_state["nu"].copy_(rs["nu"])
        state.opt_state["count"] = state.step = int(rs["opt_count"])
        if "sample_rng" in rs:
            sample_rng = rs["sample_rng"]
        if "py_random" in rs:
            random.setstate(rs["py_random"])
            np.random.set_state(rs["np_random"])
        if per_prompt_stats is not None and rs.get("tracker") is not None:
            per_prompt_stats.load_state_dict(rs["tracker"])
        mean_rewards, std_rewards, wall = list(rs.get("mean_rewards", [])), list(rs.get("std_rewards", [])), list(rs.get("wall", []))
        if wall:
            t_start -= wall[-1]
        start_epoch = rs["epoch"] + 1
        print(f"[ policy_gradient ] resumed from {rs['params_path']} (epoch {rs['epoch']}, {state.step} optimizer updates)")
    for epoch in range(start_epoch, args.num_train_epochs):
        samples = []
        for i in range(args.num_sample_batches_per_epoch):
            # ----------------------------- make prompts ----------------------------- #
            sample_prompts, training_prompts, prompt_metadata = dp.make_prompts(
                args.prompt_fn, n_devices * args.sample_batch_size, args.identical_batch, evaluate=args.evaluate, **args.prompt_kwargs)
            # ----------------------------- sample ----------------------------- #
            sample_rng, sample_seed = prng.split(sample_rng)
            sample_seeds = prng.split(sample_seed, dp.n_key_devices)
            sample_prompt_ids = pipeline.prepare_inputs(sample_prompts)
            sample_prompt_embeds = text_encode(sample_prompt_ids)
            timer()
            final_latents, latents, next_latents, log_probs, ts = pipeline(
                sample_prompt_embeds, sample_uncond_prompt_embeds, {"unet": state.params, "scheduler": sampling_scheduler_params},
                dp.sample_key(sample_seeds), args.n_inference_steps, jit=True, height=args.resolution, width=args.resolution,
                guidance_scale=args.guidance_scale, eta=args.eta)
            # ----------------------------- decode latents ----------------------------- #
            images = vae.decode(final_latents).cpu().numpy()
            print(f"[ sample ] epoch {epoch} batch {i}: {len(sample_prompts)} images in {timer():.2f}s")
            # ----------------------------- evaluate callbacks ----------------------------- #
            callbacks = executor.submit(training.evaluate_callbacks, callback_fns, images, sample_prompts, prompt_metadata)
            time.sleep(0)
            samples.append({"prompts": np.array(sample_prompts), "embeds": sample_prompt_embeds, "latents": latents,
                            "next_latents": next_latents, "log_probs": log_probs, "ts": ts, "callbacks": callbacks})
            from PIL import Image
            Image.fromarray((images[0] * 255).round().astype(np.uint8)).save(
                utils.fs.join_and_create(localpath, f"samples/{worker_id}_{epoch}_{i}.png"))

        # wait for callbacks to finish
        for sample in samples:
            sample["rewards"], sample["callback_info"] = sample.pop("callbacks").result()[args.filter_field]
        host = {k: np.concatenate([s[k] for s in samples]) for k in ("prompts", "rewards")}
        callback_info = {k: np.concatenate([np.atleast_1d(s["callback_info"][k]) for s in samples]) for k in samples[0]["callback_info"]}
        devs = {k: torch.cat([s[k] for s in samples]) for k in ("embeds", "latents", "next_latents", "log_probs", "ts")}

        # allgather rewards (for multi-process training)
        rewards, prompts = dp.gather_rewards(host["rewards"], host["prompts"].tolist(), args.num_sample_batches_per_epoch)
        if per_prompt_stats is not None:
            advantages = per_prompt_stats.update(prompts, rewards)
            if worker_id == 0:
                np.save(utils.fs.join_and_create(localpath, f"per_prompt_stats/{worker_id}_{epoch}.npy"), per_prompt_stats.get_stats())
        else:
            advantages = (rewards - np.mean(rewards)) / np.std(rewards)
        advantages = dp.local_advantages(advantages)
        print(f"mean reward: {np.mean(rewards):.4f}")
        mean_rewards.append(float(np.mean(rewards)))
        std_rewards.append(float(np.std(rewards)))
        wall.append(time.time() - t_start)

        # save data for future analysis
        np.save(utils.fs.join_and_create(localpath, f"rewards/{worker_id}_{epoch}.npy"), host["rewards"])
        np.save(utils.fs.join_and_create(localpath, f"prompts/{worker_id}_{epoch}.npy"), host["prompts"])
        np.save(utils.fs.join_and_create(localpath, f"callback_info/{worker_id}_{epoch}.npy"), callback_info)
        devs["advantages"] = torch.as_tensor(np.asarray(advantages, dtype=np.float32).reshape(-1)).to(dev)
        devs = dp.gather_global(devs, args.num_sample_batches_per_epoch)      # single_host: the per-epoch trajectory exchange

        for inner_epoch in range(args.num_inner_epochs):
            total_batch_size, num_timesteps = devs["log_probs"].shape
            assert total_batch_size == args.num_sample_batches_per_epoch * n_devices * args.sample_batch_size * (n_workers if dp.single_host else 1)
            assert num_timesteps == args.n_inference_steps
            # shuffle samples along the batch dimension, then along time independently for each sample; keep this rank's rows
            if args.num_inner_epochs == 1:
                mine = dp.shuffled_rows(devs, args.train_batch_size)          # same draws, only this rank's rows gathered
                if dp.single_host:
                    devs = None                                               # the gathered global copy is not needed again this epoch
            else:
                devs = dp.shuffle(devs)
                mine = dp.my_rows(devs, args.train_batch_size)
            total_batch_size = mine["log_probs"].shape[0]
            num_train_ts = int(num_timesteps * args.train_timestep_ratio)
            n_mini = total_batch_size // (n_devices * args.train_batch_size)
            all_infos = []
            do_opt_update = False
            train_fuse = train_fuse_default(args.train_batch_size * (2 if args.train_cfg else 1),
                                            mine["latents"].shape[-1] * mine["latents"].shape[-2])
            t_train = time.time()
            for i in range(n_mini):
                sl = slice(i * args.train_batch_size, (i + 1) * args.train_batch_size)
                for j0 in (range(0, num_train_ts, train_fuse) if train_fuse > 1 else ()):
                    # DDPO_TRAIN_FUSE=k (default 16): k consecutive timesteps of this mini-batch (same parameters: the optimizer only
                    # steps at the last timestep) as one U-Net forward/backward over k micro-batches (train_steps_fused)
                    js = range(j0, min(j0 + train_fuse, num_train_ts))
                    batches = [{"prompt_embeds": mine["embeds"][sl], "uncond_embeds": train_uncond_prompt_embeds,
                                "advantages": mine["advantages"][sl], "latents": mine["latents"][sl, j],
                                "next_latents": mine["next_latents"][sl, j], "log_probs": mine["log_probs"][sl, j],
                                "ts": mine["ts"][sl, j]} for j in js]
                    do_opt_update = (js[-1] == num_train_ts - 1) and ((i + 1) % args.train_accumulation_steps == 0)
                    if do_opt_update:
                        print(f"opt update at {i}, {js[-1]}")
                    state, infos_k = train_steps_fused(state, batches, noise_scheduler_state, pipeline.scheduler, args.train_cfg,
                                                       args.guidance_scale, args.eta, args.ppo_clip_range, do_opt_update)
                    all_infos += [torch.stack([info["approx_kl"], info["clipfrac"], info["loss"]]) for info in infos_k]
                for j in (range(num_train_ts) if train_fuse <= 1 else ()):
                    batch = {"prompt_embeds": mine["embeds"][sl], "uncond_embeds": train_uncond_prompt_embeds,
                             "advantages": mine["advantages"][sl], "latents": mine["latents"][sl, j],
                             "next_latents": mine["next_latents"][sl, j], "log_probs": mine["log_probs"][sl, j],
                             "ts": mine["ts"][sl, j]}
                    # update at the last timestep of a sequence once enough samples are accumulated
                    do_opt_update = (j == num_train_ts - 1) and ((i + 1) % args.train_accumulation_steps == 0)
                    if do_opt_update:
                        print(f"opt update at {i}, {j}")
                    state, info = train_step(state, batch, noise_scheduler_state, pipeline.scheduler, args.train_cfg,
                                             args.guidance_scale, args.eta, args.ppo_clip_range, do_opt_update)
                    all_infos.append(torch.stack([info["approx_kl"], info["clipfrac"], info["loss"]]))
            assert do_opt_update
            infos = torch.stack(all_infos).cpu().numpy()            # ONE host sync per inner epoch
            if n_workers > 1:
                infos = D.allgather_array(infos[None]).mean(0)       # lax.pmean(info)
            all_infos = {"approx_kl": infos[:, 0], "clipfrac": infos[:, 1], "loss": infos[:, 2]}
            print(f"mean info: { {k: float(v.mean()) for k, v in all_infos.items()} } | "
                  f"{len(infos)} train steps in {time.time() - t_train:.2f}s")
            if worker_id == 0:
                np.save(utils.fs.join_and_create(localpath, f"train_info/{worker_id}_{epoch}_{inner_epoch}.npy"), all_infos)

        if (epoch + 1) % args.save_freq == 0 or epoch == args.num_train_epochs - 1:
            save_rank_resume(os.path.join(args.savepath, "checkpoints"), epoch, worker_id,
                             {"sample_rng": sample_rng, "py_random": random.getstate(), "np_random": np.random.get_state()})
            if worker_id == 0:
                resume = {"epoch": epoch, "opt_count": state.opt_state["count"], "mu": state.opt_state["mu"].cpu(),
                          "nu": state.opt_state["nu"].cpu(), "sample_rng": sample_rng,
                          "tracker": None if per_prompt_stats is None else per_prompt_stats.state_dict(),
                          "mean_rewards": list(mean_rewards), "std_rewards": list(std_rewards), "wall": list(wall)}
                save_checkpoint(os.path.join(args.savepath, "checkpoints"), state.params, step=epoch, resume_state=resume,
                                synthetic_weights=pipeline.synthetic_weights)
            D.barrier()

        if worker_id == 0:
            np.save(os.path.join(localpath, "reward_vs_wallclock.npy"),
                    np.stack([np.array(wall), np.array(mean_rewards), np.array(std_rewards)], 1))
            try:
                import matplotlib
                matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                plt.clf()
                plt.plot(mean_rewards, color="black")
                m, s = np.array(mean_rewards), np.array(std_rewards)
                plt.fill_between(range(len(m)), m - s, m + s, alpha=0.4, color="blue")
                plt.savefig(os.path.join(localpath, f"log_{worker_id}.png"))
            except ImportError:
                pass
    executor.shutdown()
    return {"mean_rewards": mean_rewards, "localpath": localpath, "state": state}


if __name__ == "__main__":
    main()
