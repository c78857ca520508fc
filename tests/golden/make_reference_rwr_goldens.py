#!/usr/bin/env python
"""Generates tests/golden/reference_rwr.npz by EXECUTING the reference's RWR train step.

`python tests/golden/make_reference_rwr_goldens.py` (build container only; reads /root/reference read-only).
/root/reference/ddpo/training/diffusion.py is exec'd UNMODIFIED with numpy stand-ins for jax / jax.numpy (tests/golden/_jax_shim.py;
jax.random = the Threefry restatement pinned to the JAX documentation values, plus `randint`, an unpinned restatement) and calls
`train_step` with stubs for what it receives from outside: a closed-form U-Net (`state.apply_fn`), a deterministic text encoder, an
optimizer that records the gradient tree it is handed.  `jax.value_and_grad` is a VALUE-ONLY stand-in (no autodiff here): the fixture
pins the key tree, the posterior sample / noise / timesteps / noisy latents the U-Net is called with, the CFG mix and the loss with
and without weights.  Third-party pieces restated in this file (not reference code): diffusers 0.12.1
`vae_flax.FlaxDiagonalGaussianDistribution` and `FlaxDDPMScheduler.add_noise`."""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "reference_rwr.npz")
F = np.float32


class Arr(np.ndarray):
    """ndarray whose mean() accepts `axis=range(...)` like jax arrays do (diffusion.py:83)."""

    def mean(self, axis=None, **kw):
        if isinstance(axis, range):
            axis = tuple(axis)
        return np.asarray(self).mean(axis=axis, dtype=F, **kw).view(Arr)


def arr(x):
    return np.asarray(x, dtype=F).view(Arr)


def toy_unet(lat, t, ctx, scale, bias):
    lat = np.asarray(lat, dtype=F)
    c = np.asarray(ctx, dtype=F).mean(axis=(1, 2), dtype=F)
    tt = np.asarray(t).astype(F) / F(1000.0)
    return ((F(0.6) * lat / (F(1.0) + F(0.25) * lat * lat) + F(0.3) * tt[:, None, None, None] + F(0.5) * c[:, None, None, None]) * scale + bias).astype(F)


def main():
    import _jax_shim as SH
    from oracle import prng as OP
    jnp = SH.make_jnp()
    jax = SH.make_jax(jnp)
    jax.random.normal = lambda key, shape=(), dtype=F: arr(OP.normal(np.asarray(key, dtype=np.uint32), tuple(shape)))
    jax.random.randint = lambda key, shape, minval, maxval, dtype=np.int32: OP.randint(np.asarray(key, dtype=np.uint32), tuple(shape), minval, maxval)
    jnp.transpose = lambda x, axes=None: arr(np.transpose(np.asarray(x), axes))

    def value_and_grad(fn):
        def g(params):
            return fn(params), {k: np.zeros_like(np.asarray(v)) for k, v in params.items()}        # value only (see the docstring)
        return g
    jax.value_and_grad = value_and_grad
    sys.modules.update({"jax": jax, "jax.numpy": jnp, "jax.random": jax.random, "jax.lax": jax.lax})

    # ---- third party, restated: diffusers 0.12.1 models/vae_flax.py FlaxDiagonalGaussianDistribution
    class FlaxDiagonalGaussianDistribution:
        def __init__(self, parameters, deterministic=False):
            self.mean, self.logvar = np.split(np.asarray(parameters, dtype=F), 2, axis=-1)
            self.logvar = np.clip(self.logvar, F(-30.0), F(20.0))
            self.std = np.exp(F(0.5) * self.logvar).astype(F)

        def sample(self, key):
            return arr(self.mean + self.std * np.asarray(jax.random.normal(key, self.mean.shape)))
    vae_flax = types.ModuleType("diffusers.models.vae_flax")
    vae_flax.FlaxDiagonalGaussianDistribution = FlaxDiagonalGaussianDistribution
    models = types.ModuleType("diffusers.models")
    models.vae_flax = vae_flax
    diffusers = types.ModuleType("diffusers")
    diffusers.models = models
    sys.modules.update({"diffusers": diffusers, "diffusers.models": models, "diffusers.models.vae_flax": vae_flax})

    spec = importlib.util.spec_from_file_location("ref_diffusion", os.path.join(REF, "ddpo/training/diffusion.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)

    from oracle.diffusion import ddpm_alphas_cumprod
    acp = ddpm_alphas_cumprod()

    class NoiseScheduler:                     # ---- third party, restated: FlaxDDPMScheduler.add_noise (add_noise_common)
        config = types.SimpleNamespace(num_train_timesteps=1000)

        def __init__(self):
            self.calls = []

        def add_noise(self, state, original_samples, noise, timesteps):
            a = acp[np.asarray(timesteps)]
            sa = (a ** F(0.5)).reshape(-1, 1, 1, 1)
            sb = ((F(1.0) - a) ** F(0.5)).reshape(-1, 1, 1, 1)
            self.calls.append((np.asarray(original_samples).copy(), np.asarray(noise).copy(), np.asarray(timesteps).copy()))
            return arr(sa * np.asarray(original_samples) + sb * np.asarray(noise))

    def text_encoder(ids, params=None, train=False):
        ids = np.asarray(ids)
        base = np.linspace(-1.0, 1.0, 16, dtype=F)
        return (((ids[..., None].astype(F) % F(97.0)) / F(97.0) - F(0.5)) * params["gain"] + base[None, None, :]).astype(F),

    out = {"alphas_cumprod": acp}
    r = np.random.RandomState(5)
    for case, (B, train_cfg, g, use_w, seed) in enumerate([(4, True, 2.0, False, 0), (4, True, 5.0, True, 1), (3, False, 1.0, True, 2), (5, False, 1.0, False, 3)]):
        sched = NoiseScheduler()
        unet_calls, received = [], []
        params = {"scale": F(0.9), "bias": F(0.05)}

        def apply_fn(variables, lat, ts, ctx, train=True):
            p = variables["params"]
            unet_calls.append((np.asarray(lat).copy(), np.asarray(ts).copy(), np.asarray(ctx).copy()))
            return types.SimpleNamespace(sample=arr(toy_unet(lat, ts, ctx, p["scale"], p["bias"])))

        class State:
            def __init__(self):
                self.apply_fn, self.params = apply_fn, params

            def apply_gradients(self, grads):
                received.append(grads)
                return self
        moments = np.concatenate([r.randn(B, 8, 8, 4) * 0.8, r.randn(B, 8, 8, 4) * 0.5 - 1.0], axis=-1).astype(F)
        moments[0, 0, 0, 4] = 40.0          # logvar beyond the +20 clip
        moments[0, 0, 1, 4] = -50.0         # ... and below -30
        batch = {"vae": moments, "input_ids": r.randint(0, 49408, size=(B, 77)), "uncond_text": np.full((B, 77), 49407)}
        weights = (np.abs(r.randn(B)) / B).astype(F) if use_w else None
        rng = OP.PRNGKey(100 + seed)
        new_state, loss, new_rng = D.train_step(State(), {"gain": F(0.7)}, batch, rng, None, (sched, text_encoder, train_cfg, g), weights=weights)
        tag = f"case{case}"
        out[tag + "/moments"], out[tag + "/input_ids"], out[tag + "/uncond_text"] = moments, batch["input_ids"], batch["uncond_text"]
        out[tag + "/weights"] = weights if use_w else np.zeros(0, dtype=F)
        out[tag + "/cfg"] = np.array([int(train_cfg), g], dtype=F)
        out[tag + "/rng"], out[tag + "/new_rng"] = np.asarray(rng, dtype=np.uint32), np.asarray(new_rng, dtype=np.uint32)
        lat, noise, ts = sched.calls[0]
        out[tag + "/latents"], out[tag + "/noise"], out[tag + "/timesteps"] = lat.astype(F), noise.astype(F), ts.astype(np.int32)
        out[tag + "/noisy_latents"] = unet_calls[0][0].astype(F)
        out[tag + "/cond_embeds"] = unet_calls[0][2].astype(F)
        assert len(unet_calls) == (2 if train_cfg else 1) and len(received) == 1
        if train_cfg:
            assert np.array_equal(unet_calls[1][0], unet_calls[0][0])
            out[tag + "/uncond_embeds"] = unet_calls[1][2].astype(F)
        out[tag + "/loss"] = np.asarray(loss, dtype=F)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
