"""Oracle (test infrastructure only): PPO-clip loss on stored log-probs + gradient w.r.t. the U-Net outputs.

Restates /root/reference/ddpo/training/policy_gradient.py:60 (ADV_CLIP_MAX), :95-105 (CFG mix),
:110-118 (scoring-mode scheduler.step), :121-125 (PPO-clip), :128-134 (info).
Two forms: `loss_and_info_torch` (differentiable, any float dtype — the ground truth) and the
closed form the HIP kernel implements; tests check they agree.
"""
import math
import numpy as np
import torch

ADV_CLIP_MAX = 10.0


def _coeff_tensors(ddim, state, ts, eta, dtype):
    a_t, a_p, b_t, std = ddim.coefficients(state, np.asarray(ts), eta)
    f = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32)).to(dtype).reshape(-1, 1, 1, 1)
    return f(a_t), f(a_p), f(b_t), f(std)


def log_prob_torch(ddim, state, model_output, ts, sample, prev_sample, eta, dtype=torch.float32):
    """scheduling_ddim_flax.py:279-359 in scoring mode (prev_sample given), differentiable in model_output."""
    a_t, a_p, b_t, std = _coeff_tensors(ddim, state, ts, eta, dtype)
    sample = sample.to(dtype)
    prev_sample = prev_sample.to(dtype)
    e = model_output
    if ddim.prediction_type == "epsilon":
        x0 = (sample - b_t ** 0.5 * e) / a_t ** 0.5
    elif ddim.prediction_type == "v_prediction":
        x0 = a_t ** 0.5 * sample - b_t ** 0.5 * e
        e = a_t ** 0.5 * e + b_t ** 0.5 * sample
    elif ddim.prediction_type == "sample":
        x0 = e
    else:
        raise ValueError
    mean = a_p ** 0.5 * x0 + (1 - a_p - std ** 2) ** 0.5 * e
    std_c = torch.clamp(std, min=1e-6)
    lp = -((prev_sample.detach() - mean) ** 2) / (2 * std_c ** 2) - torch.log(std_c) - math.log(math.sqrt(2 * math.pi))
    return lp.flatten(1).mean(1)


def loss_and_info_torch(ddim, state, eps_cond, eps_uncond, batch, guidance_scale, eta, clip_range,
                        train_cfg=True, dtype=torch.float32):
    """compute_loss of policy_gradient.py:86-136 downstream of the two U-Net applies."""
    if train_cfg:
        noise_pred = eps_uncond + guidance_scale * (eps_cond - eps_uncond)
    else:
        noise_pred = eps_cond
    log_prob = log_prob_torch(ddim, state, noise_pred, batch["ts"], batch["latents"], batch["next_latents"], eta, dtype)
    adv = torch.clamp(batch["advantages"].to(dtype), -ADV_CLIP_MAX, ADV_CLIP_MAX)
    old = batch["log_probs"].to(dtype)
    ratio = torch.exp(log_prob - old)
    unclipped = -adv * ratio
    clipped = -adv * torch.clamp(ratio, 1.0 - clip_range, 1.0 + clip_range)
    loss = torch.mean(torch.maximum(unclipped, clipped))
    info = {
        "approx_kl": 0.5 * torch.mean((log_prob - old) ** 2),
        "clipfrac": torch.mean((torch.abs(ratio - 1.0) > clip_range).to(dtype)),
        "loss": loss,
    }
    return loss, info, log_prob


def closed_form_numpy(ddim, state, eps_cond, eps_uncond, latents, next_latents, ts, old_log_probs,
                      advantages, guidance_scale, eta, clip_range, train_cfg=True):
    """The algebra the fused HIP kernel implements, fp32 numpy.
    Returns loss, info dict, log_prob (B,), d_eps_cond, d_eps_uncond."""
    F = np.float32
    B = latents.shape[0]
    chw = int(np.prod(latents.shape[1:]))
    a_t, a_p, b_t, std = ddim.coefficients(state, np.asarray(ts), eta)
    r4 = lambda v: np.asarray(v, dtype=F).reshape(B, 1, 1, 1)
    a_t, a_p, b_t, std = r4(a_t), r4(a_p), r4(b_t), r4(std)
    g = F(guidance_scale)
    e = (eps_uncond + g * (eps_cond - eps_uncond)).astype(F) if train_cfg else eps_cond.astype(F)
    sq = lambda v: np.sqrt(v).astype(F)
    dirc = sq(F(1) - a_p - std ** 2)
    if ddim.prediction_type == "epsilon":
        mean = sq(a_p) * (latents - sq(b_t) * e) / sq(a_t) + dirc * e
        dmu_de = dirc - sq(a_p) * sq(b_t) / sq(a_t)
    elif ddim.prediction_type == "v_prediction":
        x0 = sq(a_t) * latents - sq(b_t) * e
        e2 = sq(a_t) * e + sq(b_t) * latents
        mean = sq(a_p) * x0 + dirc * e2
        dmu_de = dirc * sq(a_t) - sq(a_p) * sq(b_t)
    else:
        raise ValueError
    std_c = np.maximum(std, F(1e-6))
    diff = (next_latents - mean).astype(F)
    lp = (-(diff ** 2) / (F(2) * std_c ** 2) - np.log(std_c) - F(math.log(math.sqrt(2 * math.pi)))).astype(F)
    log_prob = lp.reshape(B, -1).mean(1, dtype=F)
    adv = np.clip(np.asarray(advantages, dtype=F), -ADV_CLIP_MAX, ADV_CLIP_MAX)
    ratio = np.exp(log_prob - np.asarray(old_log_probs, dtype=F)).astype(F)
    unclipped = -adv * ratio
    clipped = -adv * np.clip(ratio, F(1 - clip_range), F(1 + clip_range))
    loss = np.mean(np.maximum(unclipped, clipped), dtype=F)
    # d loss / d log_prob_b: the unclipped branch is active unless the clipped one is strictly larger
    use_unclipped = unclipped >= clipped
    dl_dlp = np.where(use_unclipped, -adv * ratio / F(B), F(0)).astype(F)
    dlp_dmu = diff / (std_c ** 2 * F(chw))
    d_e = (r4(dl_dlp) * dlp_dmu * dmu_de).astype(F)
    if train_cfg:
        d_c, d_u = (g * d_e).astype(F), ((F(1) - g) * d_e).astype(F)
    else:
        d_c, d_u = d_e, np.zeros_like(d_e)
    info = {
        "approx_kl": F(0.5) * np.mean((log_prob - old_log_probs) ** 2, dtype=F),
        "clipfrac": np.mean((np.abs(ratio - 1.0) > clip_range).astype(F), dtype=F),
        "loss": loss,
    }
    return loss, info, log_prob, d_c, d_u
