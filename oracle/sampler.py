"""Oracle (test infrastructure only): the DDPO sampling loop and PPO train step, CPU restatement.

sample():     /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:163-270 (one device)
train_step(): /root/reference/ddpo/training/policy_gradient.py:63-146 (one device, torch autograd through the
              oracle U-Net; accumulation + optimizer in oracle/optim.py)
"""
import numpy as np
import torch

from . import prng, ppo
from .ddim import DDIMOracle
from .unet import unet_forward


def sample(unet_params, cfg, ddim: DDIMOracle, sched_state, prompt_embeds, neg_embeds, key, num_inference_steps,
           height, width, guidance_scale, eta, dtype=torch.float32, unet_fn=None):
    """Returns numpy (final_latents, latents (B,T,..), next_latents (B,T,..), log_probs (B,T), ts (B,T)).
    `unet_fn(latents (2B,C,h,w) f32 numpy, t (2B,) int32 numpy, context (2B,77,D) f32 numpy) -> numpy` replaces the U-Net
    (tests/test_reference_ddim_goldens.py drives the loop with the closed-form model of the reference-run fixture)."""
    B = prompt_embeds.shape[0]
    context = torch.cat([neg_embeds, prompt_embeds]).to(dtype)
    shape = (B, cfg.in_channels, height // 8, width // 8)
    rng, seed = prng.split(np.asarray(key, dtype=np.uint32))
    latents = prng.normal(seed, shape)
    state = ddim.set_timesteps(sched_state, num_inference_steps)
    latents = latents * state.init_noise_sigma
    rng, seed = prng.split(rng)
    rng = seed
    lat_t, next_t, lp_t, ts_t = [], [], [], []
    for step in range(num_inference_steps):
        t = int(state.timesteps[step])
        if unet_fn is not None:
            noise_pred = np.asarray(unet_fn(np.concatenate([latents] * 2), np.full((2 * B,), t, dtype=np.int32),
                                            context.to(torch.float32).numpy()), dtype=np.float32)
        else:
            inp = torch.from_numpy(np.concatenate([latents] * 2)).to(dtype)
            with torch.no_grad():
                noise_pred = unet_forward(unet_params, cfg, inp, torch.full((2 * B,), t, dtype=torch.int32), context)
            noise_pred = noise_pred.to(torch.float32).numpy()
        nu, nt = noise_pred[:B], noise_pred[B:]
        guided = (nu + np.float32(guidance_scale) * (nt - nu)).astype(np.float32)
        rng, k = prng.split(rng)
        z = prng.normal(k, shape)
        new_latents, log_prob = ddim.step(state, guided, t, latents, noise=z, eta=eta)
        lat_t.append(latents); next_t.append(new_latents); lp_t.append(log_prob); ts_t.append(t)
        latents = new_latents
    ts = np.broadcast_to(np.asarray(ts_t, dtype=np.int32), (B, num_inference_steps))
    return (latents, np.stack(lat_t, 1), np.stack(next_t, 1), np.stack(lp_t, 1), ts)


def train_step_grads(unet_params, cfg, ddim: DDIMOracle, sched_state, batch, guidance_scale, eta, clip_range,
                     train_cfg=True, dtype=torch.float32):
    """compute_loss + jax.grad of ddpo/training/policy_gradient.py:86-139.  Returns ({name: grad}, info dict)."""
    leaves = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in unet_params.items()}
    lat = batch["latents"].to(dtype)
    ts = batch["ts"]
    eps_c = unet_forward(leaves, cfg, lat, ts, batch["prompt_embeds"].to(dtype))
    eps_u = unet_forward(leaves, cfg, lat, ts, batch["uncond_embeds"].to(dtype)) if train_cfg else None
    loss, info, _ = ppo.loss_and_info_torch(ddim, sched_state, eps_c, eps_u, batch, guidance_scale, eta, clip_range,
                                            train_cfg, dtype)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return grads, {k: float(v) for k, v in info.items()}, (eps_c.detach(), None if eps_u is None else eps_u.detach())
