"""DDIM scheduler with a stochastic (eta) step that also returns the per-sample Gaussian log-prob.

Host-side mirror of the reference's `FlaxDDIMScheduler` (same method names, argument meaning and errors):
  /root/reference/ddpo/diffusers_patch/scheduling_ddim_flax.py
    DDIMSchedulerState :38-61, create_state :144-170, scale_model_input :172-187, set_timesteps :189-211,
    _get_variance :213-227, step :229-361.
The per-element arithmetic of `step` runs in the fused HIP kernels `ddpo_ddim_step_fwd` (sampling mode, `key`
given) and `ddpo_ddim_logprob_ppo_fwd_bwd` (scoring mode, `prev_sample` given); this file only owns the integer
timestep bookkeeping and the ᾱ table.
"""
from dataclasses import dataclass, replace
from typing import Optional

import numpy as np
import torch

from .. import lib as L
from ..utils import prng


@dataclass(frozen=True)
class CommonSchedulerState:
    """diffusers' CommonSchedulerState: alphas / betas / alphas_cumprod in float32 (host + device copies)."""
    alphas: np.ndarray
    betas: np.ndarray
    alphas_cumprod: np.ndarray
    alphas_cumprod_dev: Optional[torch.Tensor] = None


@dataclass(frozen=True)
class DDIMSchedulerState:
    common: CommonSchedulerState
    final_alpha_cumprod: np.float32
    init_noise_sigma: np.float32
    timesteps: np.ndarray
    num_inference_steps: Optional[int] = None

    def replace(self, **kw):
        return replace(self, **kw)


class DDIMScheduler:
    """Constructor arguments and defaults follow FlaxDDIMScheduler.__init__ (:120-142)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.beta_start, self.beta_end = beta_start, beta_end
        self.beta_schedule = beta_schedule
        self.trained_betas = trained_betas
        self.set_alpha_to_one = set_alpha_to_one
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type
        if prediction_type not in L.PRED_TYPES:
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")

    # a tiny stand-in for `scheduler.config.<field>` used by pipeline/policy_gradient.py:107-116
    @property
    def config(self):
        return self

    def _common(self, device):
        f = np.float32
        T = self.num_train_timesteps
        if self.trained_betas is not None:
            betas = np.asarray(self.trained_betas, dtype=f)
        elif self.beta_schedule == "linear":
            betas = np.linspace(self.beta_start, self.beta_end, T, dtype=f)
        elif self.beta_schedule == "scaled_linear":
            betas = np.linspace(f(self.beta_start) ** f(0.5), f(self.beta_end) ** f(0.5), T, dtype=f) ** 2
        else:
            raise NotImplementedError(f"beta_schedule {self.beta_schedule} is not implemented for {self.__class__.__name__}")
        betas = betas.astype(f)
        alphas = (f(1.0) - betas).astype(f)
        ac = np.cumprod(alphas, dtype=f)
        dev = torch.from_numpy(ac.copy()).to(device) if device is not None else None
        return CommonSchedulerState(alphas=alphas, betas=betas, alphas_cumprod=ac, alphas_cumprod_dev=dev)

    def create_state(self, common: Optional[CommonSchedulerState] = None, device="cuda") -> DDIMSchedulerState:
        if common is None:
            common = self._common(device)
        final_alpha_cumprod = np.float32(1.0) if self.set_alpha_to_one else common.alphas_cumprod[0]
        timesteps = np.arange(0, self.num_train_timesteps).round()[::-1]
        return DDIMSchedulerState(common=common, final_alpha_cumprod=final_alpha_cumprod,
                                  init_noise_sigma=np.float32(1.0), timesteps=timesteps)

    def scale_model_input(self, state, sample, timestep=None):
        return sample

    def set_timesteps(self, state: DDIMSchedulerState, num_inference_steps: int, shape=()) -> DDIMSchedulerState:
        step_ratio = self.num_train_timesteps // num_inference_steps
        timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1] + self.steps_offset
        return state.replace(num_inference_steps=num_inference_steps, timesteps=timesteps.astype(np.int32))

    # ------------------------------------------------------------------------------------------
    def kernel_consts(self, state: DDIMSchedulerState, eta: float):
        if state.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        return L.make_ddim_consts(state.common.alphas_cumprod_dev, self.num_train_timesteps // state.num_inference_steps,
                                  state.final_alpha_cumprod, eta, self.prediction_type)

    @staticmethod
    def _ts_tensor(timestep, B, device):
        if torch.is_tensor(timestep):
            t = timestep.to(device=device, dtype=torch.int32)
            return t.expand(B).contiguous() if t.ndim == 0 else t.contiguous()
        t = np.asarray(timestep)
        if t.ndim == 0:
            return torch.full((B,), int(t), dtype=torch.int32, device=device)
        return torch.as_tensor(t.astype(np.int32), device=device)

    def step(self, state, model_output, timestep, sample, key=None, prev_sample=None, eta: float = 0.0):
        """Returns (prev_sample, state, log_prob).  `key`: 2 uint32 words (jax PRNG key) -> sampling mode;
        `prev_sample` given -> scoring mode (log-prob of prev_sample under the DDIM posterior)."""
        if prev_sample is not None and key is not None:
            raise ValueError("Cannot pass both key and prev_sample. Please make sure that either `key` or `prev_sample` stays `None`.")
        consts = self.kernel_consts(state, eta)
        B = sample.shape[0]
        ts = self._ts_tensor(timestep, B, sample.device)
        sample = sample.contiguous()
        model_output = model_output.contiguous()
        if prev_sample is None:
            if key is None:
                raise ValueError("either `key` or `prev_sample` is required")
            z = L.threefry_normal(key, tuple(sample.shape), device=sample.device)
            prev_sample, log_prob = L.ddim_step_fwd(model_output, model_output, sample, z, ts, 0.0, consts)
            return prev_sample, state, log_prob
        zeros = torch.zeros(B, dtype=torch.float32, device=sample.device)
        _, _, per_sample, _ = L.ddim_logprob_ppo_fwd_bwd(model_output, None, sample, prev_sample.contiguous(), ts, zeros, zeros,
                                                         1.0, 1.0, False, consts)
        return prev_sample, state, per_sample[:, 0].contiguous()

    def step_cfg(self, state, noise_pred_uncond, noise_pred_text, guidance_scale, timestep, sample, key, eta, out=None, z=None):
        """Sampling-mode step with the classifier-free-guidance combine fused in
        (pipeline_flax_stable_diffusion.py:226-235)."""
        consts = self.kernel_consts(state, eta)
        B = sample.shape[0]
        ts = self._ts_tensor(timestep, B, sample.device)
        z = L.threefry_normal(key, tuple(sample.shape), device=sample.device, out=z)
        return L.ddim_step_fwd(noise_pred_uncond, noise_pred_text, sample, z, ts, guidance_scale, consts, x_next=out)

    def __len__(self):
        return self.num_train_timesteps


# names the reference uses
FlaxDDIMScheduler = DDIMScheduler
split = prng.split
