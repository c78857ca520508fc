"""jax.random key bookkeeping (PRNGKey / split) on the host, integer-exact, through the C ABI.

Mirrors the key tree of the reference: /root/reference/pipeline/policy_gradient.py:51,201,244-245 and
/root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:196,232,252.
"""
import numpy as np

from .. import lib as L


def PRNGKey(seed):
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def split(key, num=2):
    """jax.random.split(key, num) -> (num, 2) uint32."""
    return L.threefry_bits_host(key, 2 * num).reshape(num, 2)


def normal(key, shape, device="cuda"):
    """jax.random.normal(key, shape, float32) on the device."""
    return L.threefry_normal(key, tuple(shape), device=device)
