"""Oracle (test infrastructure only): DDIM scheduler with per-sample Gaussian log-prob, numpy fp32.

Line-by-line restatement of /root/reference/ddpo/diffusers_patch/scheduling_ddim_flax.py:
  create_state :144-170, set_timesteps :189-211, _get_variance :213-227, step :229-361.
The ᾱ table comes from diffusers' CommonSchedulerState.create (un-vendored, diffusers==0.12.1):
betas = linspace(beta_start**0.5, beta_end**0.5, T)**2 for "scaled_linear"; alphas_cumprod =
cumprod(1 - betas), all float32.  Pinned by tests/golden/ddim_schedule.json.
"""
import numpy as np

F = np.float32


class DDIMState:
    """Mirror of DDIMSchedulerState (:38-61): just the fields `step` reads."""

    def __init__(self, alphas_cumprod, final_alpha_cumprod, timesteps=None, num_inference_steps=None):
        self.alphas_cumprod = alphas_cumprod
        self.final_alpha_cumprod = final_alpha_cumprod
        self.init_noise_sigma = F(1.0)
        self.timesteps = timesteps
        self.num_inference_steps = num_inference_steps


class DDIMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1,
                 prediction_type="epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.beta_start, self.beta_end = beta_start, beta_end
        self.beta_schedule = beta_schedule
        self.set_alpha_to_one = set_alpha_to_one
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type

    def create_state(self):
        T = self.num_train_timesteps
        if self.beta_schedule == "linear":
            betas = np.linspace(self.beta_start, self.beta_end, T, dtype=F)
        elif self.beta_schedule == "scaled_linear":
            betas = np.linspace(F(self.beta_start) ** F(0.5), F(self.beta_end) ** F(0.5), T, dtype=F) ** 2
        else:
            raise NotImplementedError(self.beta_schedule)
        alphas = (F(1.0) - betas.astype(F)).astype(F)
        ac = np.cumprod(alphas, dtype=F)                       # sequential float32 cumprod
        final = F(1.0) if self.set_alpha_to_one else ac[0]     # :154-158
        ts = np.arange(0, T)[::-1]
        return DDIMState(ac, final, ts, None)

    def set_timesteps(self, state, num_inference_steps):
        step_ratio = self.num_train_timesteps // num_inference_steps       # :201
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1] + self.steps_offset
        return DDIMState(state.alphas_cumprod, state.final_alpha_cumprod,
                         ts.astype(np.int32), num_inference_steps)

    def coefficients(self, state, timestep, eta):
        """Per-sample scalars of :279-327.  timestep: int or (B,) int array."""
        t = np.asarray(timestep, dtype=np.int64)
        prev_t = t - self.num_train_timesteps // state.num_inference_steps
        ac = state.alphas_cumprod
        a_t = ac[t].astype(F)
        a_p = np.where(prev_t >= 0, ac[np.maximum(prev_t, 0)], state.final_alpha_cumprod).astype(F)
        b_t = (F(1) - a_t).astype(F)
        var = ((F(1) - a_p) / (F(1) - a_t) * (F(1) - a_t / a_p)).astype(F)    # _get_variance
        std = (F(eta) * var ** F(0.5)).astype(F)
        return a_t, a_p, b_t, std

    def step(self, state, model_output, timestep, sample, noise=None, prev_sample=None, eta=0.0):
        """Returns (prev_sample, log_prob (B,)).  `noise` plays the role of normal(key, shape)."""
        if prev_sample is not None and noise is not None:
            raise ValueError("Cannot pass both key and prev_sample.")
        model_output = np.asarray(model_output, dtype=F)
        sample = np.asarray(sample, dtype=F)
        B = sample.shape[0]
        a_t, a_p, b_t, std = self.coefficients(state, timestep, eta)
        bc = lambda v: np.broadcast_to(np.asarray(v, dtype=F).reshape(-1, *([1] * (sample.ndim - 1))),
                                       (B,) + (1,) * (sample.ndim - 1)).astype(F)
        a_t, a_p, b_t, std = bc(a_t), bc(a_p), bc(b_t), bc(std)
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** F(0.5) * model_output) / a_t ** F(0.5)
        elif self.prediction_type == "sample":
            x0 = model_output
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** F(0.5) * sample - b_t ** F(0.5) * model_output
            model_output = a_t ** F(0.5) * model_output + b_t ** F(0.5) * sample
        else:
            raise ValueError(self.prediction_type)
        direction = (F(1) - a_p - std ** 2) ** F(0.5) * model_output
        mean = (a_p ** F(0.5) * x0 + direction).astype(F)
        if prev_sample is None:
            prev_sample = (mean + std * np.asarray(noise, dtype=F)).astype(F)
        prev_sample = np.asarray(prev_sample, dtype=F)
        std_c = np.maximum(std, F(1e-6))                                # clip AFTER the noise add (:348,351)
        lp = (-((prev_sample - mean) ** 2) / (F(2) * std_c ** 2) - np.log(std_c)
              - np.log(np.sqrt(F(2) * F(np.pi)))).astype(F)
        log_prob = lp.reshape(B, -1).mean(axis=1, dtype=F)              # mean over C,H,W (:359)
        return prev_sample, log_prob.astype(F)
