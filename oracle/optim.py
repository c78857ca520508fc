"""Oracle (test infrastructure only): gradient accumulation + optax clip_by_global_norm + adamw(mu_dtype=bf16).

Call sites restated: /root/reference/pipeline/policy_gradient.py:130-150 (optimizer chain),
/root/reference/ddpo/training/policy_gradient.py:32-48 (AccumulatingTrainState.apply_gradients).
The arithmetic lives in optax==0.1.5 / jax==0.4.8 (un-vendored; PARITY UNPINNED — restated from the
published algorithm):
  clip_by_global_norm(c): n = sqrt(sum g^2); g <- g if n < c else (g / n) * c
  scale_by_adam:  mu' = (1-b1)*g + b1*mu ; nu' = (1-b2)*g^2 + b2*nu ; t += 1
                  u = (mu'/(1-b1^t)) / (sqrt(nu'/(1-b2^t)) + eps) ; mu stored as bf16(mu')
  add_decayed_weights: u += wd * p (no mask) ; scale by -lr ; p <- p + u
JAX dtype-promotion detail: mu is stored in bfloat16 and `b1 * mu` multiplies a weakly-typed Python
float with a bf16 array, so the product is formed in bf16 (b1 itself is rounded to bf16 = 0.8984375
for 0.9) before being added to the f32 term.  `mu_decay_in_bf16=False` gives the "all-f32" reading.
"""
import numpy as np

F = np.float32


def bf16_round(x):
    """float32 -> nearest-even bfloat16, returned as float32 values."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    b = x.view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


def global_norm(grads):
    return np.sqrt(np.sum([np.sum(g.astype(np.float64) ** 2) for g in grads])).astype(F)


class AdamWBf16Mu:
    def __init__(self, lr=1e-5, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0,
                 mu_decay_in_bf16=True):
        self.lr, self.b1, self.b2, self.eps, self.wd, self.max_norm = lr, b1, b2, eps, weight_decay, max_grad_norm
        self.mu_decay_in_bf16 = mu_decay_in_bf16

    def init(self, params):
        return {"count": 0, "mu": [np.zeros_like(p, dtype=F) for p in params],   # bf16 values held in f32
                "nu": [np.zeros_like(p, dtype=F) for p in params]}

    def update(self, params, grads, state):
        """One optimizer step; returns (new_params, new_state, grad_norm_before_clip)."""
        gn = global_norm(grads)
        if gn < F(self.max_norm):
            g_clipped = [g.astype(F) for g in grads]
        else:
            g_clipped = [((g.astype(F) / gn) * F(self.max_norm)).astype(F) for g in grads]
        t = state["count"] + 1
        bc1 = F(1) - F(self.b1) ** F(t)
        bc2 = F(1) - F(self.b2) ** F(t)
        new_p, new_mu, new_nu = [], [], []
        for p, g, mu, nu in zip(params, g_clipped, state["mu"], state["nu"]):
            if self.mu_decay_in_bf16:
                decayed = bf16_round(bf16_round(np.array(self.b1, dtype=F)) * mu)
            else:
                decayed = F(self.b1) * mu
            m = (F(1 - self.b1) * g + decayed).astype(F)
            v = (F(1 - self.b2) * (g * g) + F(self.b2) * nu).astype(F)
            u = (m / bc1) / (np.sqrt(v / bc2) + F(self.eps))
            u = (u + F(self.wd) * p).astype(F)
            new_p.append((p + F(-self.lr) * u).astype(F))
            new_mu.append(bf16_round(m))
            new_nu.append(v)
        return new_p, {"count": t, "mu": new_mu, "nu": new_nu}, gn


class AccumulatingState:
    """AccumulatingTrainState (ddpo/training/policy_gradient.py:13-57) over a list of numpy params."""

    def __init__(self, params, opt):
        self.params = [p.astype(F) for p in params]
        self.opt = opt
        self.opt_state = opt.init(self.params)
        self.grad_acc = [np.zeros_like(p) for p in self.params]
        self.n_acc = 0
        self.step = 0
        self.last_grad_norm = None

    def apply_gradients(self, grads, do_update):
        if do_update:
            g = [((ga + gr) / F(self.n_acc + 1)).astype(F) for ga, gr in zip(self.grad_acc, grads)]
            self.params, self.opt_state, self.last_grad_norm = self.opt.update(self.params, g, self.opt_state)
            self.grad_acc = [np.zeros_like(p) for p in self.params]
            self.n_acc = 0
            self.step += 1
        else:
            self.grad_acc = [(ga + gr).astype(F) for ga, gr in zip(self.grad_acc, grads)]
            self.n_acc += 1
