"""CPU oracle for the DDPO hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy + torch-CPU fp32/fp64) of the reference
algorithm on the path named by BASELINE.json:north_star.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product package ``ddpo_amd`` never does (and fails loudly without its HIP library).

Parity status:
  * PRNG (Threefry-2x32, split, uniform, normal): PINNED — Random123 known-answer
    vectors and the values printed in the JAX documentation (tests/golden/prng_kat.json).
  * DDIM schedule / step / log-prob: restated line by line from
    /root/reference/ddpo/diffusers_patch/scheduling_ddim_flax.py; schedule constants
    pinned to tests/golden/ddim_schedule.json (published SD scaled-linear values).
  * PPO-clip loss, accumulation: restated from /root/reference/ddpo/training/policy_gradient.py;
    closed-form gradient cross-checked against torch float64 autograd.
  * U-Net / VAE / optax AdamW: the reference delegates these to un-vendored third-party
    packages (diffusers[flax]==0.12.1, optax==0.1.5, flax==0.6.9, jax==0.4.8 — none
    installable here, no weights on disk).  They are restated from the published
    architecture/algorithm and anchored on the exact parameter counts
    (859,520,964 / 49,490,199).  PARITY UNPINNED at those third-party boundaries:
    no golden output of the real JAX path exists.
"""
