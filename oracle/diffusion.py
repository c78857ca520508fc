"""Oracle (test infrastructure only): the RWR (reward-weighted regression) denoising train step.

Restates /root/reference/ddpo/training/diffusion.py:6-102 downstream of the models:
  :16      dropout_rng, sample_rng, new_train_rng = split(train_rng, 3)
  :19-23   posterior sample of the stored VAE moments (diffusers 0.12.1 vae_flax.FlaxDiagonalGaussianDistribution — third party,
           restated: mean, logvar = split(parameters, 2, axis=-1); logvar = clip(logvar, -30, 20); sample = mean + exp(logvar/2) * N(key)),
           NHWC -> NCHW, x 0.18215.  NB the reference draws it with `sample_rng` and THEN splits the same key for noise / timesteps.
  :26-36   noise = normal(noise_rng, latents.shape) (NCHW), timesteps = randint(timestep_rng, (B,), 0, num_train_timesteps)
  :40-45   noisy_latents = FlaxDDPMScheduler.add_noise = sqrt(acp[t]) latents + sqrt(1 - acp[t]) noise (diffusers — third party, restated)
  :66-81   noise_pred = uncond + g (cond - uncond) under train_cfg
  :83-90   loss_b = mean_chw (noise - noise_pred)^2; mean over the batch, or sum_b w_b loss_b with `weights`
Pinned by tests/golden/reference_rwr.npz: the reference file executed UNMODIFIED under the numpy jax shim
(tests/golden/make_reference_rwr_goldens.py); jax.random.randint is an unpinned restatement (oracle/prng.py)."""
import numpy as np
import torch

from . import prng
from .unet import unet_forward

SCALING = np.float32(0.18215)


def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = np.linspace(np.float32(beta_start) ** 0.5, np.float32(beta_end) ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
    return np.cumprod(np.float32(1.0) - betas, dtype=np.float32)


def prepare(vae_moments, train_rng, alphas_cumprod):
    """-> dict(latents, noise, timesteps, noisy_latents (all NCHW float32 / int32), new_train_rng)."""
    m = np.asarray(vae_moments, dtype=np.float32)                       # (B, h, w, 2C) NHWC
    dropout_rng, sample_rng, new_train_rng = prng.split(np.asarray(train_rng, dtype=np.uint32), 3)
    C = m.shape[-1] // 2
    mean, logvar = m[..., :C], np.clip(m[..., C:], np.float32(-30.0), np.float32(20.0))
    std = np.exp(np.float32(0.5) * logvar).astype(np.float32)
    lat = (mean + std * prng.normal(sample_rng, mean.shape)).astype(np.float32)
    lat = (np.transpose(lat, (0, 3, 1, 2)) * SCALING).astype(np.float32)
    noise_rng, timestep_rng = prng.split(sample_rng)
    noise = prng.normal(noise_rng, lat.shape)
    ts = prng.randint(timestep_rng, (lat.shape[0],), 0, len(alphas_cumprod))
    acp = np.asarray(alphas_cumprod, dtype=np.float32)[ts]
    sa = (acp ** np.float32(0.5)).reshape(-1, 1, 1, 1)
    sb = ((np.float32(1.0) - acp) ** np.float32(0.5)).reshape(-1, 1, 1, 1)
    noisy = (sa * lat + sb * noise).astype(np.float32)
    return dict(latents=lat, noise=noise, timesteps=ts.astype(np.int32), noisy_latents=noisy, new_train_rng=new_train_rng)


def loss_torch(eps_c, eps_u, noise, weights, guidance_scale, train_cfg):
    pred = eps_u + guidance_scale * (eps_c - eps_u) if train_cfg else eps_c
    per = ((noise - pred) ** 2).flatten(1).mean(1)
    return (per.mean() if weights is None else (per * weights).sum()), per


def train_step_grads(unet_params, cfg, vae_moments, prompt_embeds, uncond_embeds, train_rng, alphas_cumprod, weights=None,
                     train_cfg=True, guidance_scale=1.0, dtype=torch.float32):
    """compute_loss + jax.value_and_grad of diffusion.py:18-93 with the oracle U-Net.  Returns ({name: grad}, loss, prep)."""
    prep = prepare(vae_moments, train_rng, alphas_cumprod)
    leaves = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in unet_params.items()}
    x = torch.from_numpy(prep["noisy_latents"]).to(dtype)
    ts = torch.from_numpy(prep["timesteps"])
    eps_c = unet_forward(leaves, cfg, x, ts, prompt_embeds.to(dtype))
    eps_u = unet_forward(leaves, cfg, x, ts, uncond_embeds.to(dtype)) if train_cfg else None
    w = None if weights is None else torch.as_tensor(np.asarray(weights)).to(dtype)
    loss, per = loss_torch(eps_c, eps_u, torch.from_numpy(prep["noise"]).to(dtype), w, guidance_scale, train_cfg)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return grads, float(loss.detach()), dict(prep, per_sample=per.detach(), eps_c=eps_c.detach(), eps_u=None if eps_u is None else eps_u.detach())
