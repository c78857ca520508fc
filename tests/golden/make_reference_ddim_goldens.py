#!/usr/bin/env python
"""Generates tests/golden/reference_ddim_sampler.npz by EXECUTING the reference's DDIM scheduler and sampler loop.

`python tests/golden/make_reference_ddim_goldens.py` (build container only; reads /root/reference read-only).
  * /root/reference/ddpo/diffusers_patch/scheduling_ddim_flax.py is exec'd UNMODIFIED as a module, with numpy-backed
    stand-ins for jax / jax.numpy / flax.struct / the diffusers base classes injected (tests/golden/_jax_shim.py):
    FlaxDDIMScheduler.create_state / set_timesteps / step (sampling mode with a PRNG key, scoring mode with prev_sample
    and per-sample timesteps) are the reference's own code.
  * the body of FlaxStableDiffusionPipeline._generate (pipeline_flax_stable_diffusion.py:163-270) is lifted with `ast`
    (annotations stripped) and run with a closed-form toy U-Net, so CFG ordering, the key tree, lax.scan carry and the
    output layouts come from the reference's loop, not from a restatement.
Stand-in caveats: arithmetic is numpy float32 (x ** 0.5 is powf, XLA would emit sqrt: <= 1 ulp), jax.random is the
Threefry restatement pinned to the JAX docs values; beta schedules (CommonSchedulerState.create) are diffusers code
restated in the shim.  Tests compare with 2e-6 relative tolerance for that reason, integers exactly."""
import ast
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
OUT = os.path.join(HERE, "reference_ddim_sampler.npz")
F = np.float32


def toy_unet_numpy(lat, t, ctx):
    """Closed-form stand-in for the U-Net (shared recipe with tests/test_reference_ddim_goldens.py): float32 mul / add / div only, bounded."""
    lat = lat.astype(F)
    c = ctx.astype(F).mean(axis=(1, 2), dtype=F)
    tt = t.astype(F) / F(1000.0)
    return (F(0.6) * lat / (F(1.0) + F(0.25) * lat * lat) + F(0.3) * tt[:, None, None, None] + F(0.5) * c[:, None, None, None]).astype(F)


class _StripAnnotations(ast.NodeTransformer):
    def visit_FunctionDef(self, node):
        self.generic_visit(node)
        node.returns = None
        for a in node.args.args + node.args.kwonlyargs:
            a.annotation = None
        return node


def lift_method(path, cls, name, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    fn = _StripAnnotations().visit(sub)
                    mod = ast.Module(body=[fn], type_ignores=[])
                    ast.fix_missing_locations(mod)
                    exec(compile(mod, path, "exec"), ns)
                    return ns[name]
    raise KeyError(name)


def main():
    import _jax_shim
    _jax_shim.install()
    import jax
    import jax.numpy as jnp
    spec = importlib.util.spec_from_file_location("ref_sched", os.path.join(REF, "ddpo/diffusers_patch/scheduling_ddim_flax.py"))
    S = importlib.util.module_from_spec(spec)
    sys.modules["ref_sched"] = S
    spec.loader.exec_module(S)

    out = {}
    sd_kwargs = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                     set_alpha_to_one=False, steps_offset=1)
    rng = np.random.RandomState(0)
    B, C, H = 3, 4, 8
    for ptype in ("epsilon", "v_prediction"):
        sch = S.FlaxDDIMScheduler(prediction_type=ptype, **sd_kwargs)
        st0 = sch.create_state()
        out[f"{ptype}/alphas_cumprod"] = np.asarray(st0.common.alphas_cumprod, dtype=F)
        out[f"{ptype}/final_alpha_cumprod"] = np.asarray(st0.final_alpha_cumprod, dtype=F)
        for T in (4, 50):
            st = sch.set_timesteps(st0, T, (B, C, H, H))
            out[f"{ptype}/T{T}/timesteps"] = np.asarray(st.timesteps, dtype=np.int32)
            for si in sorted({0, T // 2, T - 1}):
                for eta in (1.0, 0.5):
                    tag = f"{ptype}/T{T}/s{si}/eta{eta}"
                    eps = rng.randn(B, C, H, H).astype(F)
                    x = (rng.randn(B, C, H, H) * 1.5).astype(F)
                    key = jax.random.PRNGKey(100 * T + si)
                    t = np.asarray(st.timesteps)[si]
                    prev, _, lp = sch.step(st, eps, t, x, key, None, eta)
                    out[tag + "/eps"], out[tag + "/x"], out[tag + "/key"] = eps, x, np.asarray(key, dtype=np.uint32)
                    out[tag + "/prev"], out[tag + "/logp"] = np.asarray(prev, dtype=F), np.asarray(lp, dtype=F)
                    # scoring mode, training style: per-sample timesteps, stored next latents (here: perturbed prev)
                    ts_vec = np.asarray(st.timesteps)[rng.randint(0, T, size=B)].astype(np.int32)
                    stored = (np.asarray(prev) + rng.randn(B, C, H, H).astype(F) * F(0.01)).astype(F)
                    _, _, lp2 = sch.step(st, eps, ts_vec, x, None, stored, eta)
                    out[tag + "/score_ts"], out[tag + "/score_prev"], out[tag + "/score_logp"] = ts_vec, stored, np.asarray(lp2, dtype=F)

    # ------------------------------------------------------------------ the sampler loop itself
    ns = {"jnp": jnp, "jax": jax, "FlaxDDIMScheduler": S.FlaxDDIMScheduler}
    generate = lift_method(os.path.join(REF, "ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py"),
                           "FlaxStableDiffusionPipeline", "_generate", ns)
    for ptype, T, g, eta, seed in (("epsilon", 4, 5.0, 1.0, 0), ("epsilon", 10, 7.5, 0.5, 3), ("v_prediction", 4, 5.0, 1.0, 7)):
        sch = S.FlaxDDIMScheduler(prediction_type=ptype, **sd_kwargs)
        st0 = sch.create_state()
        Bs, D = 2, 16
        r2 = np.random.RandomState(1000 + seed)
        emb = r2.randn(Bs, 77, D).astype(F)
        neg = np.broadcast_to(r2.randn(1, 77, D).astype(F), (Bs, 77, D)).copy()
        unet = types.SimpleNamespace(in_channels=4, apply=lambda variables, lat, t, encoder_hidden_states=None:
                                     types.SimpleNamespace(sample=toy_unet_numpy(np.asarray(lat), np.asarray(t), np.asarray(encoder_hidden_states))))
        self = types.SimpleNamespace(scheduler=sch, unet=unet, vae_scale_factor=8)
        key = jax.random.PRNGKey(seed)
        final, lat, nxt, lps, ts = generate(self, emb, neg, {"unet": None, "scheduler": st0}, key, T, 64, 64, g, eta)
        tag = f"generate/{ptype}_T{T}_g{g}_eta{eta}_seed{seed}"
        out[tag + "/emb"], out[tag + "/neg"] = emb, neg
        out[tag + "/final"], out[tag + "/latents"], out[tag + "/next_latents"] = (np.asarray(a, dtype=F) for a in (final, lat, nxt))
        out[tag + "/log_probs"], out[tag + "/ts"] = np.asarray(lps, dtype=F), np.asarray(ts, dtype=np.int32)
    # ------------------------------------------------------------------ train_step + AccumulatingTrainState
    # The reference module ddpo/training/policy_gradient.py is exec'd UNMODIFIED; `jax.grad` is a value-only stand-in
    # (no autodiff here), so the fixture pins what the loss closure computes — CFG combine, scoring-mode log-prob,
    # advantage clip, ratio, PPO-clip loss, approx_kl, clipfrac — and the gradient-accumulation rule.
    sys.modules["ddpo"] = types.ModuleType("ddpo")
    sys.modules["ddpo.diffusers_patch"] = types.ModuleType("ddpo.diffusers_patch")
    sys.modules["ddpo.diffusers_patch.scheduling_ddim_flax"] = S
    spec = importlib.util.spec_from_file_location("ref_pg", os.path.join(REF, "ddpo/training/policy_gradient.py"))
    PG = importlib.util.module_from_spec(spec)
    sys.modules["ref_pg"] = PG
    spec.loader.exec_module(PG)

    def apply_fn(variables, lat, ts, embeds, train=True):
        p = variables["params"]
        return types.SimpleNamespace(sample=(toy_unet_numpy(np.asarray(lat), np.asarray(ts), np.asarray(embeds)) * p["scale"] + p["bias"]).astype(F))

    r3 = np.random.RandomState(77)
    for ptype in ("epsilon", "v_prediction"):
        for train_cfg in (True, False):
            sch = S.FlaxDDIMScheduler(prediction_type=ptype, **sd_kwargs)
            T = 50
            st = sch.set_timesteps(sch.create_state(), T, (6, 4, 8, 8))
            Bt = 6
            lat = (r3.randn(Bt, 4, 8, 8) * 1.2).astype(F)
            ts_vec = np.asarray(st.timesteps)[r3.randint(0, T, size=Bt)].astype(np.int32)
            ts_vec[0] = 1                                          # last step: prev timestep < 0 -> final_alpha_cumprod
            emb = r3.randn(Bt, 77, 16).astype(F)
            unc = np.broadcast_to(r3.randn(1, 77, 16).astype(F), (Bt, 77, 16)).copy()
            params = {"scale": F(0.9), "bias": F(0.05)}
            # "sampled" next latents: one sampling-mode step of the same policy, then the stored log-probs are perturbed so
            # that ratios fall inside, above and below the clip range
            e_c = apply_fn({"params": params}, lat, ts_vec, emb).sample
            e_u = apply_fn({"params": params}, lat, ts_vec, unc).sample
            npred = (e_u + F(5.0) * (e_c - e_u)) if train_cfg else e_c
            nxt = np.stack([np.asarray(sch.step(st, npred[i:i + 1], int(ts_vec[i]), lat[i:i + 1], jax.random.PRNGKey(500 + i), None, 1.0)[0])[0]
                            for i in range(Bt)]).astype(F)
            _, _, lp_now = sch.step(st, npred, ts_vec, lat, None, nxt, 1.0)
            old_lp = (np.asarray(lp_now) + np.array([0.0, 3e-5, -6e-5, 2e-4, -3e-4, 5e-5], dtype=F)).astype(F)
            adv = np.array([0.7, -1.3, 14.0, -12.5, 0.2, -0.4], dtype=F)     # two beyond the +-10 advantage clip
            batch = {"latents": lat, "next_latents": nxt, "ts": ts_vec, "log_probs": old_lp, "advantages": adv,
                     "prompt_embeds": emb, "uncond_embeds": unc}
            received = []
            state = PG.AccumulatingTrainState.create(apply_fn=apply_fn, params=params,
                                                     tx=lambda p, g: (received.append(g), p)[1])
            new_state, info = PG.train_step(state, batch, st, sch, train_cfg, 5.0, 1.0, 1e-4, False)
            tag = f"train/{ptype}/cfg{int(train_cfg)}"
            for k, v in batch.items():
                out[f"{tag}/{k}"] = v
            out[tag + "/eps_cond"], out[tag + "/eps_uncond"] = e_c, e_u
            out[tag + "/loss"], out[tag + "/approx_kl"], out[tag + "/clipfrac"] = (np.asarray(info[k], dtype=F) for k in ("loss", "approx_kl", "clipfrac"))
            assert new_state.n_acc == 1 and not received
    # accumulation rule: 3 accumulating calls + 1 updating call; the optimizer must receive (g1+g2+g3+g4)/4, then reset
    grads_seq = [{"w": (r3.randn(5) * (i + 1)).astype(F)} for i in range(6)]
    received = []
    state = PG.AccumulatingTrainState.create(apply_fn=None, params={"w": np.zeros(5, dtype=F)},
                                             tx=lambda p, g: (received.append(g), {"w": p["w"] - g["w"]})[1])
    trace = []
    for i, g in enumerate(grads_seq):
        state = state.apply_gradients(grads=g, do_update=(i in (3, 5)))
        trace.append((state.n_acc, state.step, np.asarray(state.grad_acc["w"]).copy(), np.asarray(state.params["w"]).copy()))
    out["accum/grads"] = np.stack([g["w"] for g in grads_seq])
    out["accum/do_update"] = np.array([i in (3, 5) for i in range(6)])
    out["accum/n_acc"] = np.array([t[0] for t in trace], dtype=np.int32)
    out["accum/step"] = np.array([t[1] for t in trace], dtype=np.int32)
    out["accum/grad_acc"] = np.stack([t[2] for t in trace])
    out["accum/params"] = np.stack([t[3] for t in trace])
    out["accum/received"] = np.stack([g["w"] for g in received])

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
