"""Model loading and checkpointing for the DDPO entrypoint.

`load_unet` keeps the contract of /root/reference/ddpo/utils/serialization.py:322-371: returns (pipeline, params) with
params = {"text_encoder", "vae", "unet", "scheduler"}; the scheduler state is whatever scheduler the checkpoint ships
(the entrypoint replaces the scheduler by the DDIM one and only duck-types the state, reference quirk #4).

What `pretrained_model` may be (the reference hands it to `FlaxStableDiffusionPipeline.from_pretrained`, :336-341):
  * a local directory, or a hub id (`duongna/stable-diffusion-v1-4-flax`, the reference default) that resolves to a snapshot
    in an HF cache directory (`<cache>/models--org--name/snapshots/<rev>`, `$HF_HOME/hub`, `~/.cache/huggingface/hub`) or to
    `<cache>/<org>/<name>` / `<cache>/<name>` — there is no network here, so nothing is ever downloaded;
  * in that directory, any of three layouts:
      HF Flax      unet/diffusion_flax_model.msgpack, vae/diffusion_flax_model.msgpack   (nested Flax tree, HWIO / (in,out))
      HF PyTorch   unet/diffusion_pytorch_model.safetensors|.bin, vae/...                (OIHW / (out,in) -> converted)
      flat         unet.safetensors, vae.safetensors with Flax names (what `save_checkpoint` below writes)
    plus `tokenizer/` and `text_encoder/` (PyTorch weights, or `flax_model.msgpack` converted to the torch module).
  A directory that holds a U-Net but no VAE / tokenizer / text encoder is REFUSED (no silent mix of real and random parts).
  With no weights at all the call fails, unless DDPO_ALLOW_SYNTHETIC=1 (set by the benchmark, the tests and the tools): then
  deterministic random-init weights of the right architecture and a byte-level stand-in tokenizer are used,
  `pipeline.synthetic_weights` is True and the entrypoint records that in args.json and in every checkpoint.

`dtype` (reference :343-350 casts every parameter tree to it and computes in it): "float32" keeps fp32 parameters on the
fp32-equivalent datapaths (lib.DATAPATH: lib.SHIPPED_DATAPATH in the entrypoints, bf16x3 or exact fp32 on request); "bfloat16" rounds the
parameters to bf16 and selects the single-pass bf16 MFMA datapath with fp32 accumulation — what XLA does with bf16
parameters.  Activations between layers stay fp32 here (XLA would round them to bf16 as well): strictly more precise.

Checkpoints: the reference saves only the U-Net params every `save_freq` epochs and can never resume
(/root/reference/pipeline/policy_gradient.py:97-103,457-464); here `save_checkpoint` writes the params as safetensors
(weights-only, same content) AND in the reference's own flax-msgpack file format (`checkpoint_<epoch>`, readable by its
`flax:` load path; `utils/flax_msgpack.py`), plus an optional resume bundle (optimizer moments, step, RNG keys, stat tracker).
"""
import glob
import os
import re

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


def allow_synthetic():
    return os.environ.get("DDPO_ALLOW_SYNTHETIC", "0") == "1"


def resolve_pretrained(pretrained_model, cache="cache"):
    """Local directory for `pretrained_model` (a path or a hub id), or None.  Nothing is downloaded."""
    pm = str(pretrained_model)
    if os.path.isdir(pm):
        return pm
    roots = [r for r in (cache, os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
                         os.environ.get("HUGGINGFACE_HUB_CACHE"), os.path.expanduser("~/.cache/huggingface/hub")) if r]
    missing_rev = None                   # a wanted revision absent from one cache root may still be in a later one: raise only at the end
    for root in roots:
        repo = os.path.join(root, "models--" + pm.replace("/", "--"))
        snaps = [d for d in glob.glob(os.path.join(repo, "snapshots", "*")) if os.path.isdir(d)]
        if snaps:
            # the revision `from_pretrained(revision=None)` resolves: refs/main; DDPO_PRETRAINED_REVISION names another ref or a
            # commit hash; without a usable ref the most recently written snapshot (never "the lexicographically largest hash")
            ref = os.environ.get("DDPO_PRETRAINED_REVISION", "main")
            want = None
            ref_file = os.path.join(repo, "refs", ref)
            if os.path.isfile(ref_file):
                with open(ref_file) as f:
                    want = f.read().strip()
            elif ref != "main":
                want = ref
            for d in snaps:
                if want and os.path.basename(d) == want:
                    return d
            if want and ref != "main":
                missing_rev = missing_rev or repo
                continue
            pick = max(snaps, key=lambda d: (os.path.getmtime(d), os.path.basename(d)))      # ties broken by name: deterministic on one machine
            print(f"[ utils/serialization ] '{pm}': no usable refs/{ref} in '{repo}', using the most recently written snapshot {os.path.basename(pick)}")
            return pick
        for cand in (os.path.join(root, pm), os.path.join(root, pm.split("/")[-1])):
            if os.path.isdir(cand):
                return cand
    if missing_rev is not None:
        raise FileNotFoundError(f"revision '{os.environ.get('DDPO_PRETRAINED_REVISION')}' of '{pm}' is in none of the caches searched (first: '{missing_rev}')")
    return None


# ------------------------------------------------------------------------------------------------ layout converters
def _load_state_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


_VAE_ATTN = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out_0": "proj_attn"}


def torch_to_flax_tree(state, shapes):
    """diffusers-PyTorch state dict -> {flax_name: tensor in Flax layout} for the parameters named in `shapes`:
    `blocks.<i>.` -> `blocks_<i>.`; conv `weight` OIHW -> `kernel` HWIO; dense `weight` (out,in) -> `kernel` (in,out); norm
    `weight` -> `scale`; the VAE attention's newer `to_q/to_k/to_v/to_out.0` names -> `query/key/value/proj_attn`; a 1x1 conv
    stored where a dense layer is expected (or the reverse: proj_in / proj_out, VAE attention) is reshaped."""
    out = {}
    for key, val in state.items():
        name = re.sub(r"\.(\d+)(?=\.|$)", r"_\1", key)
        parts = name.split(".")
        if len(parts) >= 3 and parts[-3].startswith("attentions_") and parts[-2] in _VAE_ATTN and "transformer_blocks_0" not in name:
            parts[-2] = _VAE_ATTN[parts[-2]]          # VAE mid-block attention only: the U-Net's to_q / to_k live under attn1 / attn2
        leaf, stem = parts[-1], ".".join(parts[:-1])
        name = stem + "." + leaf
        t = torch.as_tensor(val).float()
        if leaf == "weight":
            if stem + ".scale" in shapes:
                out[stem + ".scale"] = t
                continue
            tgt = shapes.get(stem + ".kernel")
            if tgt is None:
                continue
            if t.dim() == 4:
                t = t.permute(2, 3, 1, 0)
            elif t.dim() == 2:
                t = t.t()
            if tuple(t.shape) != tuple(tgt) and t.numel() == int(np.prod(tgt)):
                t = t.reshape(tgt)            # (1,1,I,O) <-> (I,O)
            out[stem + ".kernel"] = t.contiguous()
        elif leaf == "bias" and name in shapes:
            out[name] = t
    return out


def flax_clip_text_to_torch(flat):
    """Flax `FlaxCLIPTextModel` params (flattened with '.') -> state dict of transformers' torch `CLIPTextModel`."""
    sd = {}
    for name, v in flat.items():
        t = torch.as_tensor(np.asarray(v)).float()
        if name.endswith(".kernel"):
            sd[name[:-len("kernel")] + "weight"] = t.t().contiguous()
        elif name.endswith(".scale") or name.endswith(".embedding"):
            sd[name.rsplit(".", 1)[0] + ".weight"] = t
        else:
            sd[name] = t
    return sd


def _find(local, sub, names):
    for n in names:
        p = os.path.join(local, sub, n) if sub else os.path.join(local, n)
        if os.path.exists(p):
            return p
    return None


def load_component(store, local, sub):
    """Fill `store` (a ParamStore with Flax names / layouts) from `<local>/<sub>/...` in any of the three layouts; returns the
    layout name, or None when the directory holds no weights for this component."""
    from .flax_msgpack import flatten, from_bytes
    p = _find(local, sub, ["diffusion_flax_model.msgpack"])
    if p:
        with open(p, "rb") as f:
            store.load_dict(flatten(from_bytes(f.read())))
        return "hf-flax"
    p = _find(local, sub, ["diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp32.safetensors", "diffusion_pytorch_model.bin"])
    if p:
        store.load_dict(torch_to_flax_tree(_load_state_file(p), store.shapes))
        return "hf-pytorch"
    p = _find(local, None, [sub + ".safetensors"])
    if p:
        store.load_dict(_load_state_file(p))
        return "flat"
    return None


def _load_safetensors_into(store, path):
    from safetensors.torch import load_file
    store.load_dict(load_file(path))


def load_unet(loadpath=None, epoch="latest", pretrained_model="duongna/stable-diffusion-v1-4-flax", dtype="float32",
              cache="cache", device="cuda", seed=0):
    from .. import lib as L
    dname = str(dtype).replace("torch.", "").replace("jnp.", "")
    if dname in ("bfloat16", "bf16"):
        bf16_params = True
        L.DATAPATH = "bf16"           # single-pass bf16 MFMA, fp32 accumulate: XLA's arithmetic with bf16 parameters
    elif dname in ("float32", "fp32", "f32"):
        bf16_params = False
        if L.DATAPATH == "bf16":      # an earlier load_unet(dtype=bfloat16) in this process: back to the fp32-class default
            L.DATAPATH = L.shipped_datapath()
    else:
        raise ValueError(f"dtype must be float32 or bfloat16 (reference config/base.py:71), got {dtype!r}")
    family = model_family(pretrained_model)
    ucfg = UNetConfig.named(family)
    vcfg = VAEConfig.named("tiny" if family.startswith("tiny") else "sd")
    unet, vae = UNet2DCondition(ucfg, device), VAEDecoder(vcfg, device)
    local = resolve_pretrained(pretrained_model, cache)
    synthetic, layout = True, None
    if local is not None:
        layout = load_component(unet.params, local, "unet")
    if layout is not None:
        missing = []
        if load_component(vae.params, local, "vae") is None:
            missing.append("vae")
        missing += [d for d in ("tokenizer", "text_encoder") if not os.path.isdir(os.path.join(local, d))]
        if missing:
            raise FileNotFoundError(f"'{local}' holds U-Net weights ({layout}) but no {', '.join(missing)}: refusing to mix real and "
                                    f"random-init components")
        synthetic = False
        print(f"[ utils/serialization ] loaded {family} U-Net + VAE from {local} ({layout} layout)")
    else:
        if not allow_synthetic():
            raise FileNotFoundError(
                f"no weights found for pretrained_model='{pretrained_model}' (looked for a local directory and for an HF cache snapshot "
                f"under cache='{cache}', $HF_HOME/hub, ~/.cache/huggingface/hub; nothing is downloaded).  Point --pretrained_model "
                f"at a directory with unet/, vae/, tokenizer/, text_encoder/ — or set DDPO_ALLOW_SYNTHETIC=1 to run on deterministic "
                f"RANDOM-INIT weights (benchmarks / tests only: such a run optimises noise)")
        print(f"[ utils/serialization ] WARNING: DDPO_ALLOW_SYNTHETIC=1 and no local weights for '{pretrained_model}' — using "
              f"deterministic random-init {family} weights; checkpoints of this run are marked synthetic")
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
    if bf16_params:                   # to_dtype(params, bfloat16) of the reference: round once, keep the fp32 container
        for store in (unet.params, vae.params):
            store.flat.copy_(store.flat.to(torch.bfloat16).to(torch.float32))
    if L.DATAPATH != "fp32":          # bf16 MFMA datapaths: pre-split the contraction weights once
        unet.params.pack_bf16()
        vae.params.pack_bf16(bwd=False)
    tokenizer = load_tokenizer(None if synthetic else local)
    text_encoder = TextEncoder(None if synthetic else local, hidden=ucfg.cross_attention_dim, device=device, seed=seed + 2)
    pred = ucfg.prediction_type
    scheduler = DDIMScheduler(prediction_type=pred, **SD_SCHEDULER)
    pipeline = StableDiffusionPipeline(unet, vae, scheduler, text_encoder=text_encoder, tokenizer=tokenizer)
    pipeline.synthetic_weights = synthetic
    pipeline.weights_source = None if synthetic else f"{local} ({layout})"
    pipeline.param_dtype = "bfloat16" if bf16_params else "float32"
    params = {"unet": unet.params, "vae": vae.params, "text_encoder": text_encoder,
              "scheduler": scheduler.create_state(device=device)}
    return pipeline, params


def load_vae_encoder(pretrained_model="duongna/stable-diffusion-v1-4-flax", cache="cache", device="cuda", seed=0):
    """The VAE ENCODER half (+ quant_conv) of the same checkpoint `load_unet` reads (`<dir>/vae`, any of its three layouts): what the
    reference's `vae` callback loads for the RWR sampler (/root/reference/ddpo/training/callbacks.py:37-41).  Refuses missing weights
    unless DDPO_ALLOW_SYNTHETIC=1 (then: deterministic random init, `encoder.synthetic_weights = True`)."""
    from ..models.vae import VAEEncoder
    from .. import lib as L
    family = model_family(pretrained_model)
    enc = VAEEncoder(VAEConfig.named("tiny" if family.startswith("tiny") else "sd"), device)
    local = resolve_pretrained(pretrained_model, cache)
    layout = load_component(enc.params, local, "vae") if local is not None else None
    if layout is None:
        if not allow_synthetic():
            raise FileNotFoundError(f"no VAE weights found for pretrained_model='{pretrained_model}' (the `vae` callback of pipeline/sample.py needs "
                                    f"the encoder); set DDPO_ALLOW_SYNTHETIC=1 for a seeded random-init encoder (benchmarks / tests only)")
        enc.params.init_synthetic(seed + 5)
    enc.synthetic_weights = layout is None
    if L.DATAPATH != "fp32":
        enc.params.pack_bf16(bwd=False)
    return enc


def latest_checkpoint(ckpt_dir):
    if not os.path.isdir(ckpt_dir):
        return None
    steps = sorted(int(f[len("checkpoint_"):-len(".safetensors")]) for f in os.listdir(ckpt_dir)
                   if f.startswith("checkpoint_") and f.endswith(".safetensors"))
    return os.path.join(ckpt_dir, f"checkpoint_{steps[-1]}.safetensors") if steps else None


def save_checkpoint(ckpt_dir, params, step, resume_state=None, flax_format=True, synthetic_weights=False):
    """Rank-0 write of the U-Net params (Flax names / layouts) as `checkpoint_<step>.safetensors` (+ resume bundle) and,
    with `flax_format`, as `checkpoint_<step>` in flax's msgpack encoding — the file the reference's
    `save_checkpoint_multiprocess(..., unreplicate(state.params), step=epoch)` writes and its `flax:` load path reads."""
    from safetensors.torch import save_file
    os.makedirs(ckpt_dir, exist_ok=True)
    formats = os.environ.get("DDPO_CKPT_FORMATS", "flax,safetensors").split(",")     # e.g. "flax" alone halves the disk use
    path = os.path.join(ckpt_dir, f"checkpoint_{step}.safetensors")
    host = {n: v.detach().cpu().contiguous() for n, v in params.views.items()}
    if "safetensors" in formats:
        save_file(host, path, metadata={"synthetic_weights": str(bool(synthetic_weights)), "format": "flax-names"})
    else:
        path = os.path.join(ckpt_dir, f"checkpoint_{step}")
    if flax_format and "flax" in formats:
        from .flax_msgpack import save_flax_checkpoint
        save_flax_checkpoint(ckpt_dir, {n: v.numpy() for n, v in host.items()}, step)
    if synthetic_weights:             # a run on random-init weights must not pass for a fine-tuned model
        with open(os.path.join(ckpt_dir, "SYNTHETIC_WEIGHTS"), "w") as f:
            f.write("this run started from deterministic random-init weights (DDPO_ALLOW_SYNTHETIC=1), not from a pretrained model\n")
    if resume_state is not None:
        torch.save(dict(resume_state, synthetic_weights=bool(synthetic_weights)), os.path.join(ckpt_dir, f"resume_{step}.pt"))
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
