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


def randint(key, shape, minval, maxval):
    """jax.random.randint(key, shape, minval, maxval) -> int32 numpy (host; integer-exact combination of two 32-bit Threefry draws,
    jax 0.4.8 `_randint`; the RWR step's timesteps: /root/reference/ddpo/training/diffusion.py:30-36)."""
    n = int(np.prod(shape)) if len(shape) else 1
    k1, k2 = split(key)
    hi, lo = L.threefry_bits_host(k1, n).astype(np.uint64), L.threefry_bits_host(k2, n).astype(np.uint64)
    span = np.uint64(max(int(maxval) - int(minval), 1))
    mult = np.uint64(2 ** 16) % span
    mult = (mult * mult) % span
    m32 = np.uint64(0xFFFFFFFF)
    off = ((((hi % span) * mult) & m32) + (lo % span)) & m32
    return (np.int64(minval) + (off % span).astype(np.int64)).astype(np.int32).reshape(shape)
