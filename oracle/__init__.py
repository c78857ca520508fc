"""CPU oracle for the DDPO hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy + torch-CPU fp32/fp64) of the reference
algorithm on the path named by BASELINE.json:north_star.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product package ``ddpo_amd`` never does (and fails loudly without its HIP library).

Parity status:
  * PRNG (Threefry-2x32, split, uniform, normal): PINNED — Random123 known-answer
    vectors and the values printed in the JAX documentation (tests/golden/prng_kat.json).
  * DDIM schedule / step / log-prob and the sampler loop (CFG ordering, key tree, scan carry, output layouts):
    PINNED against the reference's own code executed in place — tests/golden/make_reference_ddim_goldens.py exec's
    /root/reference/ddpo/diffusers_patch/scheduling_ddim_flax.py unmodified and runs the lifted body of
    FlaxStableDiffusionPipeline._generate under numpy stand-ins for jax.numpy / flax / the diffusers base classes
    (tests/golden/_jax_shim.py); tests/test_reference_ddim_goldens.py holds this oracle (CPU) and the HIP product (GPU)
    to the recorded outputs.  Schedule constants additionally pinned to tests/golden/ddim_schedule.json.
  * Stat tracker, prompt functions, config flag surface, jpeg rewards: PINNED the same way
    (tests/golden/make_reference_goldens.py -> reference_host_logic.json).
  * PPO-clip loss / info and gradient accumulation: PINNED — the reference's ddpo/training/policy_gradient.py is exec'd
    unmodified by the same generator (train_step's loss closure with a value-only jax.grad stand-in; the real
    AccumulatingTrainState over a minimal TrainState).  The closed-form gradient is cross-checked against torch float64
    autograd of that loss.
  * U-Net / VAE / optax AdamW: the reference delegates these to un-vendored third-party
    packages (diffusers[flax]==0.12.1, optax==0.1.5, flax==0.6.9, jax==0.4.8 — none
    installable here, no weights on disk).  They are restated from the published
    architecture/algorithm and anchored on the exact parameter counts
    (859,520,964 / 49,490,199).  PARITY UNPINNED at those third-party boundaries:
    no golden output of the real JAX path exists.
"""
