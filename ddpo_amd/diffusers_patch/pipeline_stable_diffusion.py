"""Stable-Diffusion sampler that takes pre-computed prompt embeddings and returns whole DDIM trajectories
with their log-probs — the sampling half of DDPO.

Host-side mirror of the reference's `FlaxStableDiffusionPipeline`:
  /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py
    prepare_inputs :148-161, _generate :163-270 (loop_body :204-241), __call__ :272-367, _p_generate :372-401.
One process drives one GPU, so the reference's leading device axis has length 1 here (it is accepted and
preserved for drop-in callers); rank r of a data-parallel job passes the r-th row of `prng_seed`.

Trajectories never leave HBM: one (T+1, B, C, h, w) buffer holds x_T .. x_0; `latents` = buf[:T] and
`next_latents` = buf[1:] are returned as (B, T, ...) views of it (the reference copies them to host numpy,
/root/reference/pipeline/policy_gradient.py:292-295).
"""
import os

import numpy as np
import torch

from .. import lib as L
from ..utils import prng
from .scheduling_ddim import DDIMScheduler


class StableDiffusionPipeline:
    def __init__(self, unet, vae, scheduler, text_encoder=None, tokenizer=None, dtype=torch.float32):
        self.unet = unet
        self.vae = vae
        self.scheduler = scheduler
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.dtype = dtype
        self.safety_checker = None
        self.vae_scale_factor = 2 ** (len(vae.cfg.block_out_channels) - 1) if vae is not None else 8

    # pipeline_flax_stable_diffusion.py:148-161
    def prepare_inputs(self, prompt):
        if not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        text_input = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length,
                                    truncation=True, return_tensors="np")
        return text_input.input_ids

    def _generate(self, prompt_embeds, neg_prompt_embeds, params, rng, num_inference_steps, height, width,
                  guidance_scale, eta, latents=None, jit=False):
        assert isinstance(self.scheduler, DDIMScheduler)
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        dev = self.unet.device
        B = prompt_embeds.shape[0]
        T = int(num_inference_steps)
        context = torch.cat([neg_prompt_embeds, prompt_embeds]).to(dev, torch.float32).contiguous()   # [uncond; cond]
        shape = (B, self.unet.cfg.in_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)

        rng = np.asarray(rng, dtype=np.uint32)
        traj = torch.empty((T + 1,) + shape, dtype=torch.float32, device=dev)
        if latents is None:
            rng, seed = prng.split(rng)
            L.threefry_normal(seed, shape, out=traj[0])
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
            traj[0].copy_(latents)

        state = self.scheduler.set_timesteps(params["scheduler"], num_inference_steps=T, shape=shape)
        if float(params["scheduler"].init_noise_sigma) != 1.0:
            traj[0].mul_(float(params["scheduler"].init_noise_sigma))

        rng, seed = prng.split(rng)
        rng = seed                                                  # the scan carry key (:252-255)
        step_keys = []
        for _ in range(T):
            rng, key = prng.split(rng)
            step_keys.append(key)

        timesteps = np.asarray(state.timesteps, dtype=np.int32)
        ts_dev = torch.from_numpy(np.repeat(timesteps[:, None], 2 * B, axis=1).copy()).to(dev)      # (T, 2B)
        log_probs = torch.empty(T, B, dtype=torch.float32, device=dev)
        consts = self.scheduler.kernel_consts(state, eta)
        z = torch.empty(shape, dtype=torch.float32, device=dev)
        lat2 = torch.empty((2 * B,) + shape[1:], dtype=torch.float32, device=dev)
        # jit=True (the reference's pmapped/jitted path): replay the U-Net as a captured HIP graph
        unet_fwd = self.unet.forward_graphed if jit else self.unet
        # the text context is the same for all T steps: project it through the cross-attention to_k / to_v once
        cache_ctx = hasattr(self.unet, "precompute_context")
        # both CFG halves see the same latents and timestep: the layers in front of the first cross-attention run once
        dedupe = cache_ctx and os.environ.get("DDPO_CFG_DUP", "1") != "0"
        # so is the time path per step (every sample is at the same timestep): embedding MLP + ResBlock projections once for all T
        cache_t = hasattr(self.unet, "precompute_timesteps") and os.environ.get("DDPO_TEMB_CACHE", "1") != "0"
        if cache_ctx:
            self.unet.precompute_context(context)
        if cache_t:
            self.unet.precompute_timesteps(timesteps)
        try:
            staged = jit and hasattr(self.unet, "forward_graphed_cfg") and os.environ.get("DDPO_STAGE_KERNEL", "1") != "0"
            for s in range(T):
                x = traj[s]
                if staged:
                    # [x; x], the step's time-projection row and the timesteps reach the graph's input buffers in ONE launch of the engine
                    noise_pred = self.unet.forward_graphed_cfg(x, s if cache_t else None, ts_dev[s], context, cfg_dup=dedupe)
                else:
                    if cache_t:
                        self.unet.select_timestep(s)
                    lat2[:B].copy_(x)                               # jnp.concatenate([old_latents] * 2)
                    lat2[B:].copy_(x)
                    noise_pred = unet_fwd(lat2, ts_dev[s], context, cfg_dup=True) if dedupe else unet_fwd(lat2, ts_dev[s], context)
                L.threefry_normal(step_keys[s], shape, out=z)
                L.ddim_step_fwd(noise_pred[:B], noise_pred[B:], x, z, ts_dev[s, :B], guidance_scale, consts,
                                x_next=traj[s + 1], logp=log_probs[s])
        finally:
            if cache_ctx:
                self.unet.release_context()
            if cache_t:
                self.unet.release_timesteps()
        final_latents = traj[T]
        ts = ts_dev[:, :B].transpose(0, 1)                          # (B, T)
        return (final_latents, traj[:T].transpose(0, 1), traj[1:].transpose(0, 1), log_probs.transpose(0, 1), ts)

    def __call__(self, prompt_embeds, neg_prompt_embeds, params, prng_seed, num_inference_steps=50, height=None, width=None,
                 guidance_scale=7.5, eta=0.0, latents=None, jit=False):
        """Same argument order as the reference (:272-285).  Inputs may carry the reference's leading device axis of
        length 1; outputs then carry it too: (final_latents, latents, next_latents, log_probs, ts)."""
        height = height or 64 * self.vae_scale_factor
        width = width or 64 * self.vae_scale_factor
        dev_axis = prompt_embeds.ndim == 4
        if dev_axis:
            if prompt_embeds.shape[0] != 1:
                raise ValueError("one process drives one GPU: the leading device axis must have length 1 "
                                 "(shard across ranks with torch.distributed instead of pmap)")
            prompt_embeds, neg_prompt_embeds = prompt_embeds[0], neg_prompt_embeds[0]
            prng_seed = np.asarray(prng_seed)[0]
            if latents is not None:
                latents = latents[0]
        to_t = lambda a: a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
        outs = self._generate(to_t(prompt_embeds), to_t(neg_prompt_embeds), params, prng_seed, num_inference_steps,
                              height, width, float(guidance_scale), float(eta), latents, jit=bool(jit))
        if dev_axis:
            outs = tuple(o.unsqueeze(0) for o in outs)
        return outs


FlaxStableDiffusionPipeline = StableDiffusionPipeline
