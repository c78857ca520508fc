#!/usr/bin/env python
"""RWR baseline, step 2: reward-weighted denoising fine-tuning — drop-in for the reference's pipeline/finetune.py.

    python pipeline/finetune.py --dataset compressed-animals-rwr [--key value ...]
    torchrun --nproc-per-node 8 pipeline/finetune.py --dataset a-animals-rwr       (one process per GPU)

Mirrors /root/reference/pipeline/finetune.py:48-219: the `train` experiment of config/base.py; the dataset `pipeline/sample.py` wrote
(stored VAE moments, prompts, rewards); weights = softmax(rewards * temperature) over the batch (`weighted_batch`) or over the dataset,
optionally per prompt (`weighted_dataset`, `per_prompt_weights`), divided by the pod batch size; per step
`ddpo_amd.training.diffusion.train_step` (posterior sample, DDPM noise, U-Net forward [cond + uncond], weighted MSE, backward,
clip + AdamW(bf16 mu) with the gradient all-reduce); checkpoints every `save_freq` epochs and at the end."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from ddpo_amd import utils
from ddpo_amd.training import diffusion, distributed as D
from ddpo_amd.training.policy_gradient import AccumulatingTrainState, AdamWConfig
from ddpo_amd.utils import bucket, prng
from ddpo_amd.utils.serialization import load_unet, save_checkpoint


class Parser(utils.Parser):
    config: str = "config.base"
    dataset: str = "consistent_imagenet"


def main(argv=None):
    worker_id, n_workers = D.init()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("pipeline/finetune.py needs a GPU: the DDPO engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from ddpo_amd import lib as L
    L.DATAPATH = L.shipped_datapath()

    # Rank mapping: every rank is a reference PROCESS with one local device — the multi-host run of the reference.  Its parser
    # offsets the seed by jax.process_index() (/root/reference/ddpo/utils/parser.py:173-178), BucketDataset.shard() gives process p the p-th
    # contiguous chunk of the dataset (/root/reference/ddpo/datasets/bucket.py:32-37) and the step key is row 0 of
    # split(PRNGKey(seed + p), n_local_devices = 1) (/root/reference/pipeline/finetune.py:134-135).
    args = Parser(argv).parse_args("train", process_index=worker_id)
    utils.init_logging("finetune", args.verbose)
    modelpath = None if args.iteration == 0 else args.modelpath

    # --------------------------------- models ---------------------------------#
    pipeline, params = load_unet(modelpath, epoch=args.load_epoch, pretrained_model=args.pretrained_model, dtype=args.dtype,
                                 cache=args.cache, device=dev)
    unet, text_encoder, tokenizer = pipeline.unet, params["text_encoder"], pipeline.tokenizer
    print(f"n unet params: {unet.params.n_params / 1e6:.3f}M")

    # -------------------------------- dataset ---------------------------------#
    worker_batch_size = args.train_batch_size * 1
    pod_batch_size = worker_batch_size * n_workers
    loadpath = args.loadpath.replace("gs://", "logs/")
    train_dataset, train_dataloader = bucket.get_bucket_loader(loadpath, tokenizer, batch_size=worker_batch_size, resolution=args.resolution,
                                                              max_train_samples=args.max_train_samples, host_id=worker_id, n_hosts=n_workers)
    assert not (args.weighted_batch and args.weighted_dataset), "Cannot weight over both batch and dataset"
    if args.weighted_dataset:
        train_dataset.make_weights(args.filter_field, args.temperature, args.per_prompt_weights)

    # ------------------------------- optimizer --------------------------------#
    state = AccumulatingTrainState(unet, AdamWConfig(learning_rate=args.learning_rate, b1=args.beta1, b2=args.beta2, eps=args.epsilon,
                                                     weight_decay=args.weight_decay, max_grad_norm=args.max_grad_norm))
    noise_scheduler = diffusion.DDPMNoiseScheduler(beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000)
    noise_scheduler_state = noise_scheduler.create_state(dev)

    # -------------------------- generic training setup ------------------------#
    rng = prng.PRNGKey(args.seed)
    train_rng = prng.split(rng, 1)[0]                        # jax.random.split(rng, len(jax.local_devices())): one local device per process
    num_update_steps_per_epoch = math.ceil(len(train_dataloader))
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * num_update_steps_per_epoch
    else:
        args.num_train_epochs = math.ceil(args.max_train_steps / num_update_steps_per_epoch)
    static_broadcasted = (noise_scheduler, text_encoder, args.train_cfg, args.guidance_scale)
    print(f"[ finetune ] dataset size: {len(train_dataset)} | batch size per device: {args.train_batch_size} | total pod batch size: {pod_batch_size} | "
          f"n epochs: {args.num_train_epochs} | n optimization steps: {args.max_train_steps}")

    # -------------------------------- main loop -------------------------------#
    savepath = args.savepath.replace("gs://", "logs/")
    global_step, history = 0, []
    for epoch in range(args.num_train_epochs):
        losses = []
        for batch in train_dataloader:
            if args.weighted_batch:
                rewards = np.asarray(batch[args.filter_field], dtype=np.float64).squeeze()
                rewards = D.allgather_array(np.atleast_1d(rewards)).reshape(-1)          # the softmax runs over the pod batch (utils.softmax, pmapped)
                w_all = bucket.softmax_ref(np.atleast_1d(rewards), temperature=args.temperature)
                weights = w_all[worker_id * worker_batch_size:(worker_id + 1) * worker_batch_size] if n_workers > 1 else w_all
            elif args.weighted_dataset:
                weights = np.asarray(batch["weights"], dtype=np.float64).reshape(-1) / pod_batch_size     # expected batch sum of 1
            else:
                weights = None
            state, loss, train_rng = diffusion.train_step(state, text_encoder, batch, train_rng, noise_scheduler_state, static_broadcasted,
                                                         weights=None if weights is None else np.asarray(weights, dtype=np.float32))
            losses.append(loss)
            global_step += 1
            if global_step >= args.max_train_steps:
                break
        # the reference's step returns lax.pmean(loss, "batch") — the mean over ALL devices of the pod (ddpo/training/diffusion.py:97)
        loss_avg = D.pmean_info({"loss": torch.stack(losses).mean()})["loss"] if losses else float("nan")
        history.append(loss_avg)
        if worker_id == 0:
            print(f"[ finetune ] Epoch {epoch} | steps {global_step} | average loss {loss_avg:.6f} | cfg: {args.train_cfg} | scale: {args.guidance_scale}")
        if (epoch + 1) % args.save_freq == 0 or epoch == args.num_train_epochs - 1:
            if worker_id == 0:
                save_checkpoint(os.path.join(savepath, "checkpoints"), unet.params, (epoch + 1) // args.save_freq * args.save_freq,
                                synthetic_weights=bool(pipeline.synthetic_weights))
        if global_step >= args.max_train_steps:
            break
    return history


if __name__ == "__main__":
    main()
