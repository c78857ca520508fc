"""Which reference run a rank-per-GPU job reproduces: `DDPO_DP_SEMANTICS=multi_host` (default) or `single_host`.

The reference has two ways of using N accelerators:

* multi-host, one local device per process — every process seeds `seed + process_index`
  (/root/reference/ddpo/utils/parser.py:174-179), draws ITS OWN prompts and permutations from its own `random` / numpy
  streams and only exchanges rewards / prompts (pipeline/policy_gradient.py:323-332).  `multi_host` is that run with
  process_count = world size, and it is what the entrypoint always did.
* single host, N local devices (the TPU-v3-8 / 8-GPU-node run of the paper) — ONE process, ONE `random` stream that draws the
  prompts of the whole `n_devices * sample_batch_size` batch (:235-241), noise keys `split(sample_seed, n_devices)[d]`
  (:244-245), ONE numpy stream that permutes the GLOBAL batch (so trajectories change device between sampling and training,
  :385-393) and `reshape(-1, n_devices, train_batch_size)` hands device d its rows of every mini-batch (:396-404).
  `single_host` reproduces that run seed for seed with one process per GPU: every rank seeds identically, draws the global
  prompt list and the global permutations from identically seeded streams and keeps its slice; the trajectories are
  all-gathered (one exchange per epoch, 3.3 MB per sample at 512^2 / 50 steps — RCCL over xGMI) before the global shuffle.

Everything here is host logic on numpy / torch tensors of any device; the CPU tests drive it on gloo (world 2) and compare
with a one-process run of the reference's own statements with n_devices = 2.
"""
import os

import numpy as np
import torch

from . import distributed as D
from .prompts import make_prompts as _make_prompts

MODES = ("multi_host", "single_host")


def mode_from_env():
    m = os.environ.get("DDPO_DP_SEMANTICS", "multi_host")
    if m not in MODES:
        raise ValueError(f"DDPO_DP_SEMANTICS={m!r}: expected one of {MODES}")
    return m


class DataParallel:
    """Routing of prompts, noise keys, rewards and trajectories between the ranks for one of the two reference semantics."""

    def __init__(self, mode=None, rank=None, world=None):
        self.mode = mode_from_env() if mode is None else mode
        if self.mode not in MODES:
            raise ValueError(self.mode)
        self.rank = D.process_index() if rank is None else rank
        self.world = D.process_count() if world is None else world

    @property
    def single_host(self):
        return self.mode == "single_host"

    # ------------------------------------------------------------------ seeds / keys
    @property
    def seed_process_index(self):
        """Offset added to `args.seed` (parser.py:177): the rank when every rank is a reference PROCESS, 0 when it is a DEVICE."""
        return 0 if self.single_host else self.rank

    @property
    def n_key_devices(self):
        """`n_devices` of `jax.random.split(sample_seed, n_devices)` (pipeline/policy_gradient.py:245)."""
        return self.world if self.single_host else 1

    def sample_key(self, sample_seeds):
        """This rank's row of `sample_seeds = split(sample_seed, n_key_devices)`."""
        return sample_seeds[self.rank if self.single_host else 0]

    # ------------------------------------------------------------------ prompts
    def make_prompts(self, prompt_fn, per_rank_batch, identical_batch=False, **kwargs):
        """This rank's (sample_prompts, training_prompts, prompt_metadata).  single_host: every rank draws the prompts of all
        `world * per_rank_batch` samples from its (identically seeded) `random` stream — device d owns block d, the `shard`
        reshape of ddpo/utils/preprocessing.py:35-49 — so the streams stay in lockstep on all ranks."""
        if not self.single_host:
            return _make_prompts(prompt_fn, per_rank_batch, identical_batch, **kwargs)
        p, t, m = _make_prompts(prompt_fn, self.world * per_rank_batch, identical_batch, **kwargs)
        sl = slice(self.rank * per_rank_batch, (self.rank + 1) * per_rank_batch)
        return list(p[sl]), list(t[sl]), list(m[sl])

    # ------------------------------------------------------------------ global order
    def _global_order(self, n_batches, per_rank_batch):
        """all_gather concatenates rank-major [rank][batch][sample]; the single-host run holds batch-major [batch][device][sample]
        (each sample batch is unsharded, then the batches are concatenated, :296-305,:318-321).  Returns the index array that
        turns the first order into the second."""
        idx = np.arange(self.world * n_batches * per_rank_batch).reshape(self.world, n_batches, per_rank_batch)
        return idx.transpose(1, 0, 2).reshape(-1)

    def gather_rewards(self, rewards, prompts, n_batches):
        """Global (rewards, prompts) on every rank, in the order the reproduced run holds them (what the per-prompt tracker and the
        global normalisation see: its deques depend on it)."""
        rewards = np.asarray(rewards)
        all_r = D.allgather_array(rewards)
        all_p = np.array(D.allgather_strings(list(prompts)))
        if self.single_host and self.world > 1:
            order = self._global_order(n_batches, len(rewards) // n_batches)
            all_r, all_p = all_r[order], all_p[order]
        return all_r, all_p

    def local_advantages(self, advantages):
        """multi_host: `advantages.reshape(process_count, -1)[worker_id]` (:349).  single_host: process_count is 1 — the global array
        stays whole and is routed together with the trajectories by `training_view`."""
        return np.asarray(advantages) if self.single_host else D.local_slice(advantages, self.rank, self.world)

    # ------------------------------------------------------------------ shuffle + rebatch
    def gather_global(self, devs, n_batches=1):
        """single_host: all-gather every per-sample tensor of this epoch (trajectories, log-probs, timesteps, embeddings) into the
        batch-major global order of the one-process run — the exchange step of this mode, once per epoch.  `devs['advantages']`
        is already global there (every rank computed the same array).  multi_host: nothing moves."""
        if not self.single_host or self.world == 1:
            return devs
        local = devs["log_probs"].shape[0]
        order = torch.as_tensor(self._global_order(n_batches, local // n_batches), device=devs["log_probs"].device)
        return {k: (v if k == "advantages" else D.allgather_tensor(v)[order]) for k, v in devs.items()}

    @staticmethod
    def shuffle(devs, np_random=np.random):
        """pipeline/policy_gradient.py:385-393: one permutation of the batch, then one permutation of time per sample, drawn from
        the numpy stream in that order; applied on the tensors' device."""
        total, T = devs["log_probs"].shape
        dev = devs["log_probs"].device
        perm = torch.as_tensor(np_random.permutation(total), device=dev)
        out = {k: v[perm] for k, v in devs.items()}
        perms = torch.as_tensor(np.array([np_random.permutation(T) for _ in range(total)]), device=dev)
        rows = torch.arange(total, device=dev)[:, None]
        for k in ("latents", "next_latents", "log_probs", "ts"):
            out[k] = out[k][rows, perms]
        return out

    def shuffled_rows(self, devs, train_batch_size, np_random=np.random):
        """`my_rows(shuffle(devs))` without materialising the shuffled GLOBAL set: the numpy stream is drawn exactly as `shuffle` draws it (one
        batch permutation, then one time permutation per sample of the whole set), but only this rank's rows are gathered.  single_host with
        num_inner_epochs == 1 (the default) uses it so that a rank holds the gathered global copy plus its own rows, not two global copies
        (an inner epoch after the first shuffles the already shuffled set, which needs the full copy: `shuffle` + `my_rows`)."""
        total, T = devs["log_probs"].shape
        dev = devs["log_probs"].device
        perm = np_random.permutation(total)
        perms = np.array([np_random.permutation(T) for _ in range(total)])
        if self.single_host and self.world > 1:
            assert total % (self.world * train_batch_size) == 0
            mine = np.arange(total).reshape(-1, self.world, train_batch_size)[:, self.rank].reshape(-1)
        else:
            mine = np.arange(total)
        idx = torch.as_tensor(perm[mine], device=dev)
        out = {k: v[idx] for k, v in devs.items()}
        tperm = torch.as_tensor(perms[mine], device=dev)
        rows = torch.arange(len(mine), device=dev)[:, None]
        for k in ("latents", "next_latents", "log_probs", "ts"):
            out[k] = out[k][rows, tperm]
        return out

    def my_rows(self, devs, train_batch_size):
        """`x.reshape(-1, n_devices, train_batch_size, ...)[:, d]` (:396-404): the rows device d trains on, in mini-batch order.
        multi_host: every row is mine."""
        if not self.single_host or self.world == 1:
            return devs
        total = devs["log_probs"].shape[0]
        assert total % (self.world * train_batch_size) == 0
        mine = np.arange(total).reshape(-1, self.world, train_batch_size)[:, self.rank].reshape(-1)
        idx = torch.as_tensor(mine, device=devs["log_probs"].device)
        return {k: v[idx] for k, v in devs.items()}
