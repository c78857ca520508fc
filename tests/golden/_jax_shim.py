"""numpy-backed stand-ins for the few jax / flax / diffusers names the reference's DDIM scheduler and sampler loop touch.

TEST INFRASTRUCTURE for tests/golden/make_reference_goldens.py only: with these modules injected into sys.modules the
reference's own ddpo/diffusers_patch/scheduling_ddim_flax.py can be exec'd unmodified and the body of
FlaxStableDiffusionPipeline._generate lifted and run, so the fixtures come from the reference's code, not from a
restatement.  Semantics mimicked: default float dtype float32 / int32 (jax without x64), Python scalars are weakly typed
(numpy 2 NEP-50 does the same), lax.scan as a Python loop, jax.random.{split,normal} = the Threefry restatement that is
pinned bit-for-bit to the JAX documentation values (oracle/prng.py, tests/golden/prng_kat.json).
What is NOT reference code here (third-party boundaries the reference imports from diffusers 0.12.1 / flax):
CommonSchedulerState.create (beta schedules), broadcast_to_shape_from_left, register_to_config, struct.dataclass.
"""
import dataclasses
import enum
import inspect
import sys
import types

import numpy as np

F32 = np.float32


def _f32(x):
    x = np.asarray(x)
    if x.dtype == np.float64:
        x = x.astype(F32)
    elif x.dtype == np.int64:
        x = x.astype(np.int32)
    return x


def _wrap(fn):
    def w(*a, **k):
        return _f32(fn(*[(_f32(v) if isinstance(v, (float, int, np.ndarray, np.generic)) and not isinstance(v, bool) else v) for v in a], **k))
    return w


def make_jnp():
    jnp = types.ModuleType("jax.numpy")
    jnp.float32, jnp.int32, jnp.pi = np.float32, np.int32, np.pi
    jnp.ndarray, jnp.dtype = np.ndarray, np.dtype

    def array(x, dtype=None):
        a = np.array(x, dtype=dtype) if dtype is not None else _f32(np.array(x))
        return a
    jnp.array = array
    jnp.asarray = array
    jnp.arange = lambda *a, **k: _f32(np.arange(*a, **k))
    jnp.where = lambda c, a, b: _f32(np.where(c, _f32(a), _f32(b)))
    jnp.clip = lambda x, a_min=None, a_max=None: _f32(np.clip(_f32(x), None if a_min is None else F32(a_min), None if a_max is None else F32(a_max)))
    for name in ("sqrt", "log", "exp", "mean", "sum", "concatenate", "broadcast_to", "stack", "transpose", "cumprod", "linspace",
                 "abs", "maximum", "minimum", "zeros", "ones", "zeros_like", "add"):
        setattr(jnp, name, _wrap(getattr(np, name)))
    jnp.split = lambda x, n, axis=0: np.split(x, n, axis=axis)
    return jnp


def make_jax(jnp):
    from oracle import prng as OP
    jax = types.ModuleType("jax")
    jax.numpy = jnp
    rnd = types.ModuleType("jax.random")
    rnd.KeyArray = np.ndarray
    rnd.PRNGKey = OP.PRNGKey
    rnd.split = lambda key, num=2: OP.split(np.asarray(key, dtype=np.uint32), num)
    rnd.normal = lambda key, shape=(), dtype=np.float32: OP.normal(np.asarray(key, dtype=np.uint32), tuple(shape)).astype(dtype)
    jax.random = rnd
    lax = types.ModuleType("jax.lax")
    lax.stop_gradient = lambda x: x

    def scan(f, init, xs):
        carry, ys = init, []
        for x in xs:
            carry, y = f(carry, x)
            ys.append(y)
        stacked = tuple(np.stack([np.asarray(y[i]) for y in ys]) for i in range(len(ys[0])))
        return carry, stacked
    lax.scan = scan
    lax.pmean = lambda x, axis_name=None: x                     # one device
    jax.lax = lax

    def tree_map(f, *trees):
        t0 = trees[0]
        if isinstance(t0, dict):
            return {k: tree_map(f, *[t[k] for t in trees]) for k in t0}
        if isinstance(t0, (list, tuple)):
            return type(t0)(tree_map(f, *[t[i] for t in trees]) for i in range(len(t0)))
        return f(*trees)
    jax.tree_map = tree_map

    def grad(fn, has_aux=False):
        """VALUE-ONLY stand-in: runs the reference's loss closure and hands back the marker gradients the caller
        planted in GRAD_MARKER (there is no autodiff here; the fixtures pin loss / info and the accumulation rule)."""
        def g(params):
            out = fn(params)
            marker = tree_map(lambda p: jax.GRAD_MARKER(p), params)
            return (marker, out[1]) if has_aux else marker
        return g
    jax.grad = grad
    jax.GRAD_MARKER = lambda p: np.zeros_like(p)
    return jax


def make_flax():
    flax = types.ModuleType("flax")
    struct = types.ModuleType("flax.struct")

    def dataclass(cls):
        cls = dataclasses.dataclass(cls)
        cls.replace = lambda self, **kw: dataclasses.replace(self, **kw)
        return cls
    struct.dataclass = dataclass
    flax.struct = struct
    training = types.ModuleType("flax.training")
    ts_mod = types.ModuleType("flax.training.train_state")

    class TrainState:
        """Minimal flax.training.train_state.TrainState: apply_gradients hands the gradients to `tx(params, grads)`."""
        _fields = ("step", "apply_fn", "params", "tx", "opt_state")

        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        def replace(self, **kw):
            new = object.__new__(type(self))
            new.__dict__.update(self.__dict__)
            new.__dict__.update(kw)
            return new

        def apply_gradients(self, *, grads, **kwargs):
            new_params = self.tx(self.params, grads)
            return self.replace(step=self.step + 1, params=new_params, **kwargs)

        @classmethod
        def create(cls, *, apply_fn, params, tx, **kwargs):
            return cls(step=0, apply_fn=apply_fn, params=params, tx=tx, opt_state=None, **kwargs)
    ts_mod.TrainState = TrainState
    core = types.ModuleType("flax.core")
    fd = types.ModuleType("flax.core.frozen_dict")

    class FrozenDict(dict):
        def __class_getitem__(cls, item):
            return cls
    fd.FrozenDict = FrozenDict
    flax.training, training.train_state, flax.core, core.frozen_dict = training, ts_mod, core, fd
    flax._extra = {"flax.training": training, "flax.training.train_state": ts_mod, "flax.core": core, "flax.core.frozen_dict": fd}
    return flax


def make_diffusers(jnp):
    diffusers = types.ModuleType("diffusers")
    cu = types.ModuleType("diffusers.configuration_utils")

    class ConfigMixin:
        def register_to_config(self, **kw):
            for k, v in kw.items():
                setattr(self.config, k, v)

    def register_to_config(init):
        sig = inspect.signature(init)

        def wrapped(self, *a, **k):
            bound = sig.bind(self, *a, **k)
            bound.apply_defaults()
            cfg = {n: v for n, v in bound.arguments.items() if n not in ("self", "kwargs")}
            self.config = types.SimpleNamespace(**cfg)
            init(self, *a, **k)
        return wrapped
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    ut = types.ModuleType("diffusers.utils")
    ut.deprecate = lambda *a, take_from=None, **k: None
    sch = types.ModuleType("diffusers.schedulers")
    su = types.ModuleType("diffusers.schedulers.scheduling_utils_flax")

    class FlaxKarrasDiffusionSchedulers(enum.Enum):
        FlaxDDIMScheduler = 1

    class FlaxSchedulerMixin:
        pass

    @dataclasses.dataclass
    class FlaxSchedulerOutput:
        prev_sample: np.ndarray

    @dataclasses.dataclass
    class CommonSchedulerState:
        alphas: np.ndarray
        betas: np.ndarray
        alphas_cumprod: np.ndarray

        @classmethod
        def create(cls, scheduler):                        # diffusers 0.12.1 scheduling_utils_flax.CommonSchedulerState.create
            c = scheduler.config
            if c.trained_betas is not None:
                betas = np.asarray(c.trained_betas, dtype=scheduler.dtype)
            elif c.beta_schedule == "linear":
                betas = np.linspace(c.beta_start, c.beta_end, c.num_train_timesteps, dtype=scheduler.dtype)
            elif c.beta_schedule == "scaled_linear":
                betas = np.linspace(c.beta_start ** 0.5, c.beta_end ** 0.5, c.num_train_timesteps, dtype=scheduler.dtype) ** 2
            else:
                raise NotImplementedError(c.beta_schedule)
            alphas = (1.0 - betas).astype(scheduler.dtype)
            return cls(alphas=alphas, betas=betas, alphas_cumprod=np.cumprod(alphas, axis=0, dtype=scheduler.dtype))

    def broadcast_to_shape_from_left(x, shape):
        x = np.asarray(x)
        assert len(shape) >= x.ndim
        return np.broadcast_to(x.reshape(x.shape + (1,) * (len(shape) - x.ndim)), shape)

    su.FlaxKarrasDiffusionSchedulers, su.FlaxSchedulerMixin, su.FlaxSchedulerOutput = FlaxKarrasDiffusionSchedulers, FlaxSchedulerMixin, FlaxSchedulerOutput
    su.CommonSchedulerState, su.broadcast_to_shape_from_left = CommonSchedulerState, broadcast_to_shape_from_left
    su.add_noise_common = su.get_velocity_common = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    diffusers.configuration_utils, diffusers.utils, diffusers.schedulers = cu, ut, sch
    sch.scheduling_utils_flax = su
    return {"diffusers": diffusers, "diffusers.configuration_utils": cu, "diffusers.utils": ut, "diffusers.schedulers": sch,
            "diffusers.schedulers.scheduling_utils_flax": su}


def install():
    """Inject the stand-ins; returns the names added so the caller can remove them again."""
    jnp = make_jnp()
    jax = make_jax(jnp)
    flax = make_flax()
    mods = {"jax": jax, "jax.numpy": jnp, "jax.random": jax.random, "jax.lax": jax.lax, "flax": flax, "flax.struct": flax.struct}
    mods.update(flax._extra)
    mods.update(make_diffusers(jnp))
    dm = types.ModuleType("diffusers.models")
    dm.vae_flax = types.ModuleType("diffusers.models.vae_flax")            # imported by the reference, never used on this path
    mods.update({"diffusers.models": dm, "diffusers.models.vae_flax": dm.vae_flax, "optax": types.ModuleType("optax")})
    sys.modules.update(mods)
    return list(mods)
