"""Model loading and checkpointing for the DDPO entrypoint.

`load_unet` keeps the contract of /root/reference/ddpo/utils/serialization.py:322-371: returns (pipeline, params) with
params = {"text_encoder", "vae", "unet", "scheduler"}; the scheduler state is whatever scheduler the checkpoint ships
(the entrypoint replaces the scheduler by the DDIM one and only duck-types the state, reference quirk #4).

Weight sources, in order:
  1. a local directory `pretrained_model` holding `unet.safetensors` / `vae.safetensors` with Flax-named tensors in
     Flax layouts (what `save_checkpoint` below writes; a converted HF Flax checkpoint has the same names);
  2. otherwise deterministic random-init weights of the right architecture (this container has no network and no
     checkpoints) — a warning is printed and `pipeline.synthetic_weights` is True.
Checkpoints: the reference saves only the U-Net params every `save_freq` epochs and can never resume
(/root/reference/pipeline/policy_gradient.py:97-103,457-464); here `save_checkpoint` writes the params as safetensors
(weights-only, same content) AND in the reference's own flax-msgpack file format (`checkpoint_<epoch>`, readable by its
`flax:` load path; `utils/flax_msgpack.py`), plus an optional resume bundle (optimizer moments, step, RNG keys, stat tracker).
"""
import os

import numpy as np
import torch

from ..models.unet import UNet2DCondition, UNetConfig
from ..models.vae import VAEDecoder, VAEConfig
from ..models.text import TextEncoder, load_tokenizer
from ..diffusers_patch.scheduling_ddim import DDIMScheduler
from ..diffusers_patch.pipeline_stable_diffusion import StableDiffusionPipeline

SD_SCHEDULER = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                    trained_betas=None, set_alpha_to_one=False, steps_offset=1)


def model_family(pretrained_model):
    forced = os.environ.get("DDPO_MODEL_CONFIG")
    if forced:
        return forced
    name = str(pretrained_model).lower()
    return "sd21" if ("stable-diffusion-2" in name or "sd21" in name or "sd-2" in name) else "sd15"


def _load_safetensors_into(store, path):
    from safetensors.torch import load_file
    store.load_dict(load_file(path))


def load_unet(loadpath=None, epoch="latest", pretrained_model="duongna/stable-diffusion-v1-4-flax", dtype="float32",
              cache="cache", device="cuda", seed=0):
    if dtype not in ("float32", torch.float32):
        raise NotImplementedError("this engine computes on the exact-fp32 MFMA datapath; dtype must be float32")
    family = model_family(pretrained_model)
    ucfg = UNetConfig.named(family)
    vcfg = VAEConfig.named("tiny" if family.startswith("tiny") else "sd")
    unet, vae = UNet2DCondition(ucfg, device), VAEDecoder(vcfg, device)
    local = pretrained_model if os.path.isdir(str(pretrained_model)) else None
    synthetic = True
    if local and os.path.exists(os.path.join(local, "unet.safetensors")):
        _load_safetensors_into(unet.params, os.path.join(local, "unet.safetensors"))
        _load_safetensors_into(vae.params, os.path.join(local, "vae.safetensors"))
        synthetic = False
    else:
        print(f"[ utils/serialization ] WARNING: no local weights for '{pretrained_model}' (offline) — using deterministic "
              f"random-init {family} weights")
        unet.params.init_synthetic(seed)
        vae.params.init_synthetic(seed + 1)
    if loadpath and str(loadpath).startswith("flax:"):
        # the reference's own checkpoint files (flax msgpack of the U-Net param tree), reference :357-362
        from .flax_msgpack import load_flax_checkpoint
        path = str(loadpath)[len("flax:"):]
        print(f"[ utils/serialization ] Loading flax checkpoint from {path}")
        unet.params.load_dict(load_flax_checkpoint(path))
    elif loadpath:
        ck = latest_checkpoint(loadpath) if epoch == "latest" else os.path.join(loadpath, f"checkpoint_{epoch}.safetensors")
        if ck and os.path.exists(ck):
            print(f"[ utils/serialization ] loading fine-tuned U-Net from {ck}")
            _load_safetensors_into(unet.params, ck)
        else:                      # only the flax-format file was written (DDPO_CKPT_FORMATS=flax), or a reference run's directory
            from .flax_msgpack import load_flax_checkpoint
            fx = loadpath if epoch == "latest" else os.path.join(loadpath, f"checkpoint_{epoch}")
            print(f"[ utils/serialization ] loading fine-tuned U-Net from flax checkpoint {fx}")
            unet.params.load_dict(load_flax_checkpoint(fx))
    from .. import lib as L
    if L.DATAPATH != "fp32":          # bf16-split MFMA datapath: pre-split the contraction weights once
        unet.params.pack_bf16()
        vae.params.pack_bf16(bwd=False)
    tokenizer = load_tokenizer(local)
    text_encoder = TextEncoder(local, hidden=ucfg.cross_attention_dim, device=device, seed=seed + 2)
    pred = ucfg.prediction_type
    scheduler = DDIMScheduler(prediction_type=pred, **SD_SCHEDULER)
    pipeline = StableDiffusionPipeline(unet, vae, scheduler, text_encoder=text_encoder, tokenizer=tokenizer)
    pipeline.synthetic_weights = synthetic
    params = {"unet": unet.params, "vae": vae.params, "text_encoder": text_encoder,
              "scheduler": scheduler.create_state(device=device)}
    return pipeline, params


def latest_checkpoint(ckpt_dir):
    if not os.path.isdir(ckpt_dir):
        return None
    steps = sorted(int(f[len("checkpoint_"):-len(".safetensors")]) for f in os.listdir(ckpt_dir)
                   if f.startswith("checkpoint_") and f.endswith(".safetensors"))
    return os.path.join(ckpt_dir, f"checkpoint_{steps[-1]}.safetensors") if steps else None


def save_checkpoint(ckpt_dir, params, step, resume_state=None, flax_format=True):
    """Rank-0 write of the U-Net params (Flax names / layouts) as `checkpoint_<step>.safetensors` (+ resume bundle) and,
    with `flax_format`, as `checkpoint_<step>` in flax's msgpack encoding — the file the reference's
    `save_checkpoint_multiprocess(..., unreplicate(state.params), step=epoch)` writes and its `flax:` load path reads."""
    from safetensors.torch import save_file
    os.makedirs(ckpt_dir, exist_ok=True)
    formats = os.environ.get("DDPO_CKPT_FORMATS", "flax,safetensors").split(",")     # e.g. "flax" alone halves the disk use
    path = os.path.join(ckpt_dir, f"checkpoint_{step}.safetensors")
    host = {n: v.detach().cpu().contiguous() for n, v in params.views.items()}
    if "safetensors" in formats:
        save_file(host, path)
    else:
        path = os.path.join(ckpt_dir, f"checkpoint_{step}")
    if flax_format and "flax" in formats:
        from .flax_msgpack import save_flax_checkpoint
        save_flax_checkpoint(ckpt_dir, {n: v.numpy() for n, v in host.items()}, step)
    if resume_state is not None:
        torch.save(resume_state, os.path.join(ckpt_dir, f"resume_{step}.pt"))
    return path


def save_rank_resume(ckpt_dir, step, rank, rank_state):
    """Per-rank part of a resumable checkpoint (host RNG streams and the JAX-style sampling key of THIS rank; the shared part —
    parameters, optimizer moments, tracker, reward history — is rank 0's `resume_<step>.pt`)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    torch.save(rank_state, os.path.join(ckpt_dir, f"resume_{step}.rank{rank}.pt"))


def load_resume(ckpt_dir, rank=0, epoch=None):
    """Everything `pipeline/policy_gradient.py` needs to continue a run after epoch `epoch` (default: the latest one that has a
    resume bundle).  The reference cannot resume at all (/root/reference/pipeline/policy_gradient.py:97-103 never loads): this is
    an addition, switched on with DDPO_RESUME=<run>/checkpoints.  Returns a dict: epoch, params_path (safetensors or flax file),
    opt_count, mu, nu, tracker, mean_rewards, std_rewards, wall, and — when the rank file exists — sample_rng, py_random,
    np_random."""
    if not os.path.isdir(ckpt_dir):
        raise FileNotFoundError(f"no checkpoint directory {ckpt_dir}")
    if epoch is None:
        steps = sorted(int(f[len("resume_"):-len(".pt")]) for f in os.listdir(ckpt_dir)
                       if f.startswith("resume_") and f.endswith(".pt") and ".rank" not in f)
        if not steps:
            raise FileNotFoundError(f"no resume_<epoch>.pt in {ckpt_dir}")
        epoch = steps[-1]
    epoch = int(epoch)
    out = dict(torch.load(os.path.join(ckpt_dir, f"resume_{epoch}.pt"), map_location="cpu", weights_only=False))
    out["epoch"] = epoch
    st = os.path.join(ckpt_dir, f"checkpoint_{epoch}.safetensors")
    out["params_path"] = st if os.path.exists(st) else os.path.join(ckpt_dir, f"checkpoint_{epoch}")
    rf = os.path.join(ckpt_dir, f"resume_{epoch}.rank{rank}.pt")
    if os.path.exists(rf):
        out.update(torch.load(rf, map_location="cpu", weights_only=False))
    return out


def load_params_file(store, path):
    """Fill a ParamStore from a `checkpoint_<epoch>.safetensors` or flax-msgpack `checkpoint_<epoch>` file."""
    if path.endswith(".safetensors"):
        _load_safetensors_into(store, path)
    else:
        from .flax_msgpack import load_flax_checkpoint
        store.load_dict(load_flax_checkpoint(path))
