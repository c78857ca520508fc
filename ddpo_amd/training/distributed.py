"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" for the CPU tests).  Replaces the reference's pmap / multihost collectives:

  lax.pmean(grad)                         -> ONE all-reduce(sum) of the flat gradient buffer per optimizer update
                                             (ddpo_amd/training/policy_gradient.py), mean folded into the AdamW kernel
  lax.pmean(info)                         -> all-reduce of 3 floats, once per logging point          (pmean_info)
  multihost_utils.process_allgather(r)    -> all-gather of per-rank rewards, tiled                   (allgather_array)
  process_allgather(prompt_ids) + decode  -> all_gather_object of the prompt strings                  (allgather_strings)
  jax.process_index() / process_count()   -> rank / world size
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # DDPO_FORCE_DIST=1 builds the group for a single rank too (RCCL smoke test on a 1-GPU box)
    if (world > 1 or os.environ.get("DDPO_FORCE_DIST") == "1") and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("DDPO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return process_index(), process_count()


def process_index():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def process_count():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _through_backend():
    """True when collectives must go through torch.distributed: more than one rank, or a forced single-rank group (smoke test)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("DDPO_FORCE_DIST") == "1")


def _comm_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def allgather_array(x):
    """process_allgather(x, tiled=True): concatenate equally-shaped per-rank numpy arrays along axis 0, on every rank."""
    x = np.asarray(x)
    if not _through_backend():
        return x
    t = torch.from_numpy(np.ascontiguousarray(x)).to(_comm_device())
    out = [torch.empty_like(t) for _ in range(process_count())]
    dist.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()


def allgather_tensor(t):
    """Concatenate equally-shaped per-rank tensors along axis 0 on every rank, staying on the tensor's device when the backend can
    (nccl = RCCL: one all_gather_into_tensor over xGMI; gloo: through host memory)."""
    if not _through_backend():
        return t
    src = t.contiguous()
    cd = _comm_device()
    if src.device != cd:
        src = src.to(cd)
    out = torch.empty((process_count() * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=cd)
    dist.all_gather_into_tensor(out, src)
    return out.to(t.device)


def allgather_strings(strings):
    strings = list(strings)
    if not _through_backend():
        return strings
    out = [None] * process_count()
    dist.all_gather_object(out, strings)
    return [s for part in out for s in part]


def broadcast_object(obj, src=0):
    """Rank `src`'s (picklable) object on every rank; the object itself without a process group."""
    if not _through_backend():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def pmean_info(info):
    """Mean over ranks of a dict of scalars (device tensors or floats) with ONE small all-reduce."""
    keys = sorted(info)
    vals = torch.stack([torch.as_tensor(info[k], dtype=torch.float32).reshape(()).to(_comm_device() if _through_backend() else "cpu")
                        for k in keys])
    if _through_backend():
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        vals = vals / process_count()
    return {k: float(v) for k, v in zip(keys, vals.cpu())}


def allreduce_sum_(flat, group=None):
    """In-place sum over ranks of a flat buffer (the gradient all-reduce)."""
    if _through_backend():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class GradBucketer:
    """Bucketed gradient all-reduce that can run BEHIND the backward pass (reference: `jax.lax.pmean(grad, "batch")`,
    /root/reference/ddpo/training/policy_gradient.py:141).

    The flat gradient buffer is laid out in forward order and the backward pass finishes parameters from the END of the buffer
    towards its start, so "everything from offset `lo` on is final" describes its progress.  The buffer is cut into fixed buckets
    counted from the end; `ready(lo)` launches the all-reduce of every not-yet-launched bucket that lies entirely at or above `lo`
    on a side stream (after an event recorded on the launching stream: the reduce of a bucket starts when the kernels that produced
    it have retired, while the backward of the earlier layers keeps running); `finish()` launches what is left and makes the calling
    stream wait for all of it.  Element-wise the result is one all_reduce(SUM) of the whole buffer — bucketing changes which elements
    travel together, not what is added to what; with more than two ranks the ORDER in which a backend adds the ranks' contributions may
    depend on the segmentation (rounding-level differences, identical on every rank) (tests/test_distributed_cpu.py: gloo, worlds 2 and
    4, random progress)."""

    def __init__(self, flat, bucket_numel=64 << 20, group=None):          # 64 Mi floats = 256 MiB (DDPO_GRAD_BUCKET_MIB)
        self.flat, self.group = flat, group
        n = flat.numel()
        bn = max(1, int(bucket_numel))
        self.bounds = [(max(0, n - (i + 1) * bn), n - i * bn) for i in range((n + bn - 1) // bn)]        # from the tail
        self.next = 0                                    # index of the first bucket not yet launched
        self.works = []
        self.active = _through_backend()
        self.stream = None
        if self.active and flat.is_cuda:
            self.stream = torch.cuda.Stream(flat.device)

    def _launch(self, lo, hi):
        view = self.flat[lo:hi]
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, lo):
        """All gradients at flat offsets >= lo are final (on the calling stream)."""
        if not self.active:
            return
        while self.next < len(self.bounds) and self.bounds[self.next][0] >= lo:
            self._launch(*self.bounds[self.next])
            self.next += 1

    def finish(self):
        """Reduce whatever has not been launched; on return the calling stream is ordered behind every bucket."""
        if not self.active:
            return
        self.ready(0)
        for w in self.works:
            w.wait()
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        self.works = []


def local_slice(global_array, rank=None, world=None):
    """advantages.reshape(process_count, -1)[worker_id] (reference pipeline/policy_gradient.py:349)."""
    rank = process_index() if rank is None else rank
    world = process_count() if world is None else world
    return np.asarray(global_array).reshape(world, -1)[rank]


def barrier():
    if _through_backend():
        dist.barrier()
