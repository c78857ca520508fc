"""Oracle (test infrastructure only): JAX-compatible counter PRNG, restated in numpy.

The reference draws all sampling noise through ``jax.random`` (jax==0.4.8, default
non-partitionable Threefry-2x32 implementation — an un-vendored third-party dependency):
  * initial latents   /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:196-197
  * per-step key split ...pipeline_flax_stable_diffusion.py:232,252
  * per-step noise    /root/reference/ddpo/diffusers_patch/scheduling_ddim_flax.py:347
  * key tree root     /root/reference/pipeline/policy_gradient.py:51,201,244-245

Restated from the published algorithm (Salmon et al., "Parallel random numbers: as easy as
1, 2, 3", SC'11; jax/_src/prng.py; xla/client/lib/math.cc ErfInv32 = Giles' single-precision
polynomial).  Pinned by tests/golden/prng_kat.json (Random123 KATs + JAX documentation values).
Integer words are bit-exact; floats follow the same f32 operation sequence.
"""
import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_PARITY = np.uint32(0x1BD11BDA)


def _rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    """20-round Threefry-2x32 on uint32 arrays x0/x1 with scalar key words k0/k1."""
    with np.errstate(over="ignore"):
        k0 = np.uint32(k0)
        k1 = np.uint32(k1)
        ks = (k0, k1, k0 ^ k1 ^ _PARITY)
        x0 = np.asarray(x0, dtype=np.uint32).copy()
        x1 = np.asarray(x1, dtype=np.uint32).copy()
        x0 += ks[0]
        x1 += ks[1]
        for g in range(5):
            for r in _ROT[g % 2]:
                x0 += x1
                x1 = _rotl(x1, r)
                x1 ^= x0
            x0 += ks[(g + 1) % 3]
            x1 += ks[(g + 2) % 3] + np.uint32(g + 1)
    return x0, x1


def PRNGKey(seed):
    """jax.random.PRNGKey for a python int seed: uint32[2] = {seed >> 32, seed & 0xffffffff}."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def random_bits(key, n):
    """threefry_random_bits: counters iota(n) (zero-padded to even); the FIRST half is word 0 and the
    SECOND half word 1 of n/2 Threefry blocks; output = concat(out0, out1)[:n]."""
    n = int(n)
    cnt = np.arange(n + (n & 1), dtype=np.uint32)
    if n & 1:
        cnt[-1] = 0
    half = cnt.size // 2
    o0, o1 = threefry2x32(key[0], key[1], cnt[:half], cnt[half:])
    return np.concatenate([o0, o1])[:n]


def split(key, num=2):
    """jax.random.split: bits(key, 2*num).reshape(num, 2)."""
    return random_bits(key, 2 * num).reshape(num, 2)


def randint(key, shape, minval, maxval):
    """jax.random.randint(key, shape, minval, maxval, int32) for in-range bounds (jax 0.4.8 `_randint`): two independent 32-bit
    draws from split(key) are combined as ((hi % span) * (2^32 % span) + lo % span) % span, all in uint32 arithmetic, which makes
    the result uniform over spans that do not divide 2^32.  Restated from the published source (the RWR train step draws its
    timesteps with it: /root/reference/ddpo/training/diffusion.py:30-36); PARITY UNPINNED — no JAX here and no documented values."""
    n = int(np.prod(shape)) if len(shape) else 1
    k1, k2 = split(np.asarray(key, dtype=np.uint32))
    hi, lo = random_bits(k1, n).astype(np.uint64), random_bits(k2, n).astype(np.uint64)
    span = np.uint64(max(int(maxval) - int(minval), 1))
    mult = np.uint64(2 ** 16) % span
    mult = (mult * mult) % span
    M32 = np.uint64(0xFFFFFFFF)
    off = (((hi % span) * mult) & M32) + (lo % span)
    off = (off & M32) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32).reshape(shape)


def uniform(key, shape, minval=0.0, maxval=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    bits = random_bits(key, n)
    fbits = (bits >> np.uint32(9)) | np.uint32(0x3F800000)
    floats = fbits.view(np.float32) - np.float32(1.0)
    lo = np.float32(minval)
    hi = np.float32(maxval)
    out = floats * np.float32(hi - lo) + lo
    return np.maximum(lo, out).reshape(shape)


_W_LT5 = np.array([2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
                   -0.00125372503, -0.00417768164, 0.246640727, 1.50140941], dtype=np.float32)
_W_GE5 = np.array([-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
                   -0.0076224613, 0.00943887047, 1.00167406, 2.83297682], dtype=np.float32)


def erfinv_f32(x):
    """XLA ErfInv32 (Giles' polynomial), all arithmetic in float32."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = -np.log1p(-(x * x)).astype(np.float32)
        lt = w < np.float32(5.0)
        w2 = np.where(lt, w - np.float32(2.5), np.sqrt(w) - np.float32(3.0)).astype(np.float32)
        p = np.where(lt, _W_LT5[0], _W_GE5[0]).astype(np.float32)
        for i in range(1, 9):
            p = (np.where(lt, _W_LT5[i], _W_GE5[i]).astype(np.float32) + p * w2).astype(np.float32)
        res = (p * x).astype(np.float32)
    return np.where(np.abs(x) == 1, x * np.float32(np.inf), res).astype(np.float32)


def normal(key, shape):
    """jax.random.normal(key, shape, float32) = sqrt(2) * erfinv(uniform(nextafter(-1,0), 1))."""
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0), dtype=np.float32)
    u = uniform(key, shape, lo, 1.0)
    return (np.float32(np.sqrt(2.0)) * erfinv_f32(u)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# The reference's key tree (pipeline/policy_gradient.py:51,201,244-245; pipeline_flax...py:196-255)
# ---------------------------------------------------------------------------------------------
def sample_key_tree(seed, n_devices, n_batches):
    """Yields, for each sample batch, the per-device keys handed to the sampler."""
    rng = PRNGKey(seed)
    _train_rng, sample_rng = split(rng)
    out = []
    for _ in range(n_batches):
        sample_rng, sample_seed = split(sample_rng)
        out.append(split(sample_seed, n_devices))
    return out


def device_noise_stream(dev_key, shape, n_steps):
    """Per-device noise: (init_latents, [z_0 .. z_{T-1}]) exactly as `_generate` draws them."""
    rng, seed = split(dev_key)
    init = normal(seed, shape)
    rng, seed = split(rng)          # scan carry key
    rng = seed
    zs = []
    for _ in range(n_steps):
        rng, key = split(rng)
        zs.append(normal(key, shape))
    return init, zs
