"""world_size-2 gloo tests (CPU) of the data-parallel path: reward / prompt all-gather, per-rank advantage slices that are
identical on every rank, info pmean, and the gradient all-reduce + 1/(n_acc*world) scaling being equivalent to the
reference's per-micro-step pmean followed by accumulation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ddpo_amd.training import distributed as D
    from ddpo_amd.utils.stat_tracking import PerPromptStatTracker
    r, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(100 + rank)
    rewards = rng.standard_normal((4, 1))                    # jpeg-style (N,1) float64 rewards of this rank
    prompts = [f"p{(rank * 4 + i) % 3}" for i in range(4)]
    all_r = D.allgather_array(rewards)
    all_p = D.allgather_strings(prompts)
    tracker = PerPromptStatTracker(32, 2)
    adv = tracker.update(np.array(all_p), all_r)             # every rank computes the SAME global advantages
    mine = D.local_slice(adv, rank, world)
    info = D.pmean_info({"loss": float(rank + 1), "clipfrac": 0.5 * rank})
    # gradient path: two micro-steps of local grads, one all-reduce(sum) at the update, scale 1/(n_acc*world)
    g = [torch.from_numpy(rng.standard_normal(10)) for _ in range(2)]
    acc = g[0] + g[1]
    D.allreduce_sum_(acc)
    acc = acc / (2 * world)
    q.put((rank, all_r, all_p, adv, mine, info, acc.numpy(), [x.numpy() for x in g]))
    D.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_data_parallel_logic():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, all_r0, all_p0, adv0, mine0, info0, acc0, g0), (r1, all_r1, all_p1, adv1, mine1, info1, acc1, g1) = res
    assert np.array_equal(all_r0, all_r1) and all_p0 == all_p1 and np.array_equal(adv0, adv1)     # identical on all ranks
    assert all_r0.shape == (8, 1) and len(all_p0) == 8
    # rank r owns the r-th contiguous block, in rank order
    rng0, rng1 = np.random.default_rng(100), np.random.default_rng(101)
    np.testing.assert_array_equal(all_r0, np.concatenate([rng0.standard_normal((4, 1)), rng1.standard_normal((4, 1))]))
    np.testing.assert_array_equal(mine0, adv0.reshape(2, -1)[0])
    np.testing.assert_array_equal(mine1, adv0.reshape(2, -1)[1])
    assert info0 == info1 and info0["loss"] == pytest.approx(1.5) and info0["clipfrac"] == pytest.approx(0.25)
    # reference order: pmean over ranks at every micro-step, accumulate, divide by n_acc -> same as ours
    ref = sum(0.5 * (a + b) for a, b in zip(g0, g1)) / 2
    np.testing.assert_allclose(acc0, ref, rtol=1e-12)
    np.testing.assert_allclose(acc1, ref, rtol=1e-12)


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ddpo_amd.training import distributed as D
    D.init(backend="gloo")
    out = []
    for case, (n, bn) in enumerate([(1000, 128), (1000, 1000), (777, 50), (64, 1 << 20), (4096, 1)]):
        g = torch.from_numpy(np.random.default_rng(7 * case + rank).standard_normal(n))
        blocking = g.clone()
        dist.all_reduce(blocking, op=dist.ReduceOp.SUM)
        # the backward reports "everything from `lo` on is final" in decreasing, irregular steps (the same steps on every rank:
        # all ranks run the same backward), sometimes repeating an offset, and finish() sweeps up the rest
        steps = sorted(set(np.random.default_rng(1000 + case).integers(0, n, size=9).tolist()), reverse=True)
        b = D.GradBucketer(g, bucket_numel=bn)
        launched = []
        for lo in steps + steps[-1:]:
            b.ready(lo)
            launched.append(b.next)
        assert all(b.bounds[i][0] >= steps[-1] for i in range(b.next))        # nothing below the reported offset was touched early
        b.finish()
        assert b.next == len(b.bounds) and b.bounds[0][1] == n and b.bounds[-1][0] == 0
        assert sum(hi - lo for lo, hi in b.bounds) == n                       # the buckets partition the buffer
        # bit-identical for two ranks (a + b has one order); for more ranks the backend's reduction tree depends on how the buffer is
        # segmented, so the sums agree to rounding — and every rank must hold the SAME bits (checked by the parent through the digest)
        ok = torch.equal(g, blocking) if world == 2 else bool(torch.allclose(g, blocking, rtol=1e-13, atol=1e-13))
        out.append((ok, launched))
    q.put((rank, out))
    D.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_gradient_all_reduce_equals_the_blocking_one(world):
    """GradBucketer (the all-reduce that hangs on the backward's progress) on gloo: for every bucket size and every
    progress pattern the buffer ends equal to one blocking all_reduce(SUM) — bit for bit with two ranks, to rounding (1e-13) with four,
    where the backend's reduction order depends on the segmentation."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(o[0] for o in out), (rank, [o[0] for o in out])
    assert all(r[1][i][1] == res[0][1][i][1] for r in res for i in range(len(r[1])))      # same launch schedule on every rank


def test_backward_progress_tracker_on_the_real_parameter_layout():
    """unet.backward reports blocks by name as their kernels are queued; the tracker turns that into "every gradient at offset >= lo is
    final".  On the SD-1.5 layout, walked in backward order: offsets decrease monotonically, end at 0, and at every step all parameters
    at or above the reported offset belong to blocks already reported (also when a block's resnets and attentions interleave)."""
    import math
    from ddpo_amd.models import unet as U
    shapes = U.unet_param_shapes(U.UNetConfig.named("sd15"))
    offsets, off = {}, 0
    for n, shp in shapes.items():
        offsets[n] = off
        off += (math.prod(shp) + 3) // 4 * 4
    names = list(offsets)
    blocks = []
    for n in names:
        parts = n.split(".")
        pre = ".".join(parts[:2]) if parts[0].startswith(("down_blocks", "mid_block", "up_blocks")) else parts[0]
        if pre not in blocks:
            blocks.append(pre)
    prog = U.UNet2DCondition._Progress(offsets)
    reported, last = set(), off
    for b in reversed(blocks):
        lo = prog.report(b + ".")
        reported.add(b)
        assert lo is not None and lo <= last
        last = lo
        for n in names:
            if offsets[n] >= lo:
                parts = n.split(".")
                pre = ".".join(parts[:2]) if parts[0].startswith(("down_blocks", "mid_block", "up_blocks")) else parts[0]
                assert pre in reported, (b, n)
    assert last == 0
    # out-of-order reports never expose an unfinished block
    prog2 = U.UNet2DCondition._Progress(offsets)
    assert prog2.report("down_blocks_0.resnets_0.") is None and prog2.report("conv_out.") == offsets["conv_out.kernel"]
