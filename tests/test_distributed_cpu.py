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
