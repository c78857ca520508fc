#!/usr/bin/env python
"""RWR baseline, step 1: sample a reward-labelled dataset — drop-in for the reference's pipeline/sample.py.

    python pipeline/sample.py --dataset compressed-animals-rwr [--key value ...]
    torchrun --nproc-per-node 8 pipeline/sample.py --dataset a-animals-rwr        (one process per GPU)

Mirrors /root/reference/pipeline/sample.py:15-170 step for step: the `sample` experiment of config/base.py, the same prompt / key
streams, HIP sampler (CFG, DDIM eta) -> VAE decode -> reward callbacks [filter_field, "vae"] -> masker (percentile / streaming
percentile / threshold of the rewards) -> writer.  What differs, deliberately: the writer is a directory of `.npz` shards on the local
filesystem (ddpo_amd/utils/bucket.py) instead of HDF5 shards in a GCS bucket; `n_devices` is 1 per process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from ddpo_amd import training, utils
from ddpo_amd.models.text import make_uncond_text
from ddpo_amd.training import callbacks as CB, distributed as D
from ddpo_amd.training.callbacks import encode_jpeg
from ddpo_amd.utils import bucket, prng
from ddpo_amd.utils.serialization import load_unet, load_vae_encoder


class Parser(utils.Parser):
    config: str = "config.base"
    dataset: str = "compressed_dogs"


def main(argv=None):
    worker_id, n_workers = D.init()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("pipeline/sample.py needs a GPU: the DDPO engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from ddpo_amd import lib as L
    L.DATAPATH = L.shipped_datapath()

    args = Parser(argv).parse_args("sample", process_index=worker_id)
    rng = prng.PRNGKey(args.seed)
    n_devices = 1
    batch_size = n_devices * args.n_samples_per_device
    pod_batch_size = batch_size * n_workers
    print(f"[ sample ] local devices: {n_devices} | pod devices: {n_devices * n_workers} | worker batch_size: {batch_size} | pod batch size: {pod_batch_size}")

    # ----------------------------------- loading ----------------------------------#
    loadpath = None if args.iteration == 0 else args.loadpath
    pipeline, params = load_unet(loadpath, epoch=args.load_epoch, pretrained_model=args.pretrained_model, cache=args.cache, device=dev)
    pipeline.safety_checker = None
    CB.set_vae_encoder(load_vae_encoder(args.pretrained_model, cache=args.cache, device=dev))
    callback_keys = [args.filter_field, "vae"]
    callback_fns = {key: training.callback_fns[key]() for key in callback_keys}
    if args.guidance_scale == "auto":
        import json
        args.guidance_scale = float(json.load(open(os.path.join(args.loadpath, "metadata.json")))["guidance_scale"])
    text_encode = params["text_encoder"]

    # ----------------------------------- bucket -----------------------------------#
    savepath = args.savepath.replace("gs://", "logs/")
    run_id = D.broadcast_object(bucket.new_run_id())          # one id per sampling run: rank 0's, in every rank's shard names and manifest
    writer = bucket.LocalWriter(savepath, split_size=args.local_size, rank=worker_id, run_id=run_id)
    writer.configure("images", encode_fn=encode_jpeg, decode_fn=bucket.decode_jpeg)
    writer.configure("inference_prompts")
    writer.configure("training_prompts")
    for key in callback_fns:
        writer.configure(key)

    # -------------------------------- uncond prompt --------------------------------#
    uncond_prompt_embeds = text_encode(make_uncond_text(pipeline.tokenizer, batch_size))
    print(f"[ sample ] embed uncond prompts: {tuple(uncond_prompt_embeds.shape)}")

    # ---------------------------------- main loop ---------------------------------#
    masker = bucket.make_masker(args.mask_mode, args.mask_param)
    avg = bucket.StreamingAverage()
    timer = utils.Timer()
    print(f"[ sample ] max_samples: {args.max_samples} | max_steps: {args.max_steps} | eval: {args.evaluate}")
    n_steps, n_samples, all_rewards = 0, 0, []
    while True:
        rng, prng_seed = prng.split(rng)
        prng_seeds = prng.split(prng_seed, n_devices)
        inference_prompts, training_prompts, prompt_metadata = training.make_prompts(
            args.prompt_fn, batch_size, args.identical_batch, evaluate=args.evaluate, **args.prompt_kwargs)
        print(f"[ sample ] prompts: {inference_prompts[:2]}")
        prompt_embeds = text_encode(pipeline.prepare_inputs(inference_prompts))
        final_latents, *_ = pipeline(prompt_embeds, uncond_prompt_embeds, params, prng_seeds[0], args.n_inference_steps, jit=True,
                                     height=args.resolution, width=args.resolution, guidance_scale=args.guidance_scale, eta=args.eta)
        images = pipeline.vae.decode(final_latents).cpu().numpy().astype(np.float32)
        print(f"[ sample ] {len(images)} samples in {timer():.2f} seconds | eval: {args.evaluate}")
        infos = training.evaluate_callbacks(callback_fns, images, training_prompts, prompt_metadata)
        rewards, metadata = infos[args.filter_field]
        rewards = np.asarray(rewards)
        all_rewards.append(rewards.squeeze())
        avg(rewards.mean().item())
        mask = masker(rewards)
        print(rewards.squeeze())
        batch = {"inference_prompts": inference_prompts, "training_prompts": training_prompts, "images": images,
                 **{key: rew for key, (rew, _) in infos.items()}}
        n_added = writer.add_batch(batch, mask=mask)
        n_steps += 1
        n_samples += int(np.sum(D.allgather_array(np.asarray([n_added], dtype=np.int64))))          # utils.worker_sum
        print(f"[ sample ] batch {n_steps} / {args.max_steps} | saved: {n_added} | total: {int(n_samples)} / {args.max_samples} | "
              f"average: {avg.avg:.3f} | mask: {masker} | saving: {timer():.2f} seconds\n")
        if args.max_steps is not None and n_steps >= args.max_steps:
            break
        if args.max_samples is not None and n_samples >= args.max_samples:
            break
    writer.close(world=n_workers, metadata={"guidance_scale": args.guidance_scale, "filter_field": args.filter_field, "n_samples": int(n_samples),
                           "synthetic_weights": bool(pipeline.synthetic_weights)})
    return savepath


if __name__ == "__main__":
    main()
