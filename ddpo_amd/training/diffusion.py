"""RWR (reward-weighted regression) denoising train step on the HIP engine — the baseline the paper compares DDPO with.

Host-side mirror of /root/reference/ddpo/training/diffusion.py:6-102 (`train_step`, `vae_decode`, `text_encode`):
same argument meaning, same key tree (dropout / sample / next key from split(train_rng, 3); the posterior sample drawn with
`sample_rng`, which is then split again for noise and timesteps — the reference's own order), same loss
(`((noise - noise_pred) ** 2).mean(chw)`, batch mean or `(loss * weights).sum()`).  Device work:
  * ddpo_threefry_normal x 2 (posterior noise in NHWC order, diffusion noise in NCHW order: the shapes JAX draws them in),
  * ddpo_rwr_noisy_latents (posterior sample + NHWC -> NCHW + x 0.18215 + DDPM add_noise, one kernel),
  * the U-Net forward on [uncond; cond] as one batch (train_cfg), ddpo_rwr_mse_fwd_bwd (loss + closed-form d loss / d eps),
    the U-Net backward accumulating into the flat gradient buffer, one fused clip + AdamW(bf16 mu) update
    (`AccumulatingTrainState.apply_gradients(do_update=True)` with nothing accumulated = flax TrainState.apply_gradients,
    the all-reduce of `lax.pmean(grad)` included).
"""
import numpy as np
import torch

from .. import lib as L
from ..utils import prng
from .policy_gradient import AccumulatingTrainState

VAE_SCALING = 0.18215
NUM_TRAIN_TIMESTEPS = 1000


class DDPMNoiseScheduler:
    """What finetune.py builds (`diffusers.FlaxDDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    num_train_timesteps=1000)`, reference pipeline/finetune.py:112-119) reduced to what train_step touches: the alphas_cumprod
    table (`create_state`) and `config.num_train_timesteps`."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=NUM_TRAIN_TIMESTEPS):
        self.num_train_timesteps = int(num_train_timesteps)
        betas = np.linspace(np.float32(beta_start) ** 0.5, np.float32(beta_end) ** 0.5, self.num_train_timesteps, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod(np.float32(1.0) - betas, dtype=np.float32)

    def create_state(self, device="cuda"):
        return torch.from_numpy(self.alphas_cumprod).to(device)


def prepare_latents(vae_moments, train_rng, noise_scheduler_state):
    """reference :16-45.  vae_moments (B,h,w,2C) NHWC device tensor -> (noise, timesteps, noisy_latents, new_train_rng)."""
    dev = vae_moments.device
    B, h, w, C2 = vae_moments.shape
    C = C2 // 2
    _dropout_rng, sample_rng, new_train_rng = prng.split(np.asarray(train_rng, dtype=np.uint32), 3)
    e1 = prng.normal(sample_rng, (B, h, w, C), device=dev)               # latent_dist.sample(sample_rng): drawn in the NHWC shape
    noise_rng, timestep_rng = prng.split(sample_rng)
    noise = prng.normal(noise_rng, (B, C, h, w), device=dev)             # jax.random.normal(noise_rng, latents.shape), NCHW
    ts_host = prng.randint(timestep_rng, (B,), 0, noise_scheduler_state.numel())
    ts = torch.from_numpy(ts_host).to(dev)
    _lat, noisy = L.rwr_noisy_latents(vae_moments.contiguous(), e1, noise, ts, noise_scheduler_state, VAE_SCALING)
    return noise, ts, noisy, new_train_rng


def train_step(state: AccumulatingTrainState, text_encoder, batch, train_rng, noise_scheduler_state, static_broadcasted, weights=None):
    """One RWR step (reference :6-102).  batch: "vae" (b,h,w,8) posterior moments, "input_ids" / "uncond_text" (b,77) token ids
    (or precomputed "prompt_embeds" / "uncond_embeds"); static_broadcasted = (noise_scheduler, text_encoder, train_cfg,
    guidance_scale) as in the reference (`text_encoder` may be passed either way).  Returns (state, loss (device scalar), new_train_rng)."""
    noise_scheduler, te2, train_cfg, guidance_scale = static_broadcasted
    text_encoder = text_encoder if text_encoder is not None else te2
    unet = state.unet
    dev = unet.device
    moments = torch.as_tensor(batch["vae"], dtype=torch.float32, device=dev)
    noise, ts, noisy, new_train_rng = prepare_latents(moments, train_rng, noise_scheduler_state)
    emb = batch["prompt_embeds"] if "prompt_embeds" in batch else text_encoder(batch["input_ids"])
    b = noisy.shape[0]
    tape = []
    if train_cfg:
        unc = batch["uncond_embeds"] if "uncond_embeds" in batch else text_encoder(batch["uncond_text"])
        out = unet.forward(torch.cat([noisy, noisy]), torch.cat([ts, ts]), torch.cat([unc, emb]).contiguous(), tape=tape)
        eps_u, eps_c = out[:b].contiguous(), out[b:].contiguous()
    else:
        eps_u, eps_c = None, unet.forward(noisy, ts, emb.contiguous(), tape=tape)
    w = None if weights is None else torch.as_tensor(np.asarray(weights, dtype=np.float32).reshape(-1) if not torch.is_tensor(weights) else weights,
                                                     dtype=torch.float32, device=dev).reshape(-1).contiguous()
    if w is not None and w.numel() != b:
        raise ValueError(f"weights has {w.numel()} entries for a batch of {b} (reference: assert loss.size == weights.size)")
    d_c, d_u, per_sample, loss = L.rwr_mse_fwd_bwd(eps_c, eps_u, noise, w, float(guidance_scale), bool(train_cfg))
    unet.backward(tape, torch.cat([d_u, d_c]) if train_cfg else d_c)
    state = state.apply_gradients(do_update=True)             # pmean(grad) + optimizer step, every call (flax TrainState)
    return state, loss[0], new_train_rng


def vae_decode(latents, vae):
    """reference :96-102: latents (b,4,h,w) NCHW -> images (b,H,W,3) in [0,1] (the decoder applies 1/0.18215, /2 + 0.5, clip)."""
    return vae.decode(latents)


def text_encode(input_ids, text_encoder):
    return text_encoder(input_ids)
