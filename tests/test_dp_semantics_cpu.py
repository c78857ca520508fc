"""DDPO_DP_SEMANTICS=single_host (ddpo_amd/training/dp.py): a world-2 job of one process per device must route prompts, noise
keys, rewards / advantages and trajectories exactly like the reference's ONE-process run with n_devices = 2
(/root/reference/pipeline/policy_gradient.py:235-245, 296-349, 385-404).  The one-process side below is a restatement of
those reference statements on numpy (shard = reshape(n_dev, -1), unshard, concatenate over sample batches, perm, per-sample
time perms, reshape(-1, n_dev, train_bs)); the two-process side runs the product module on gloo with CPU tensors.
multi_host (the default) is checked against the per-process statements the entrypoint always used."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

SEED, WORLD, SBS, TBS, NB, T = 7, 2, 4, 2, 2, 5          # sample batch 4 / device, train batch 2 / device, 2 sample batches, 5 timesteps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _trajectory(global_ids):
    """Synthetic per-sample data tagged with the global sample id (so a mis-routed row is visible)."""
    g = np.asarray(global_ids, dtype=np.float32)
    lat = g[:, None, None] * 100 + np.arange(T, dtype=np.float32)[None, :, None] + np.zeros((1, 1, 3), np.float32)
    return {"embeds": g[:, None] + np.zeros((1, 2), np.float32), "latents": lat, "next_latents": lat + 0.5,
            "log_probs": g[:, None] * 10 + np.arange(T, dtype=np.float32)[None], "ts": (g[:, None] * 0 + np.arange(T)[None]).astype(np.int64)}


def _reward_of(prompt, gid):
    return float(len(prompt)) + 0.01 * gid


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      DDPO_DP_SEMANTICS=mode)
    import torch.distributed as dist
    from ddpo_amd.training import distributed as D
    from ddpo_amd.training.dp import DataParallel
    from ddpo_amd.utils import prng
    from ddpo_amd.utils.stat_tracking import PerPromptStatTracker
    D.init(backend="gloo")
    dp = DataParallel()
    assert dp.mode == mode and (dp.rank, dp.world) == (rank, world)
    seed = SEED + dp.seed_process_index
    random.seed(seed)
    np.random.seed(seed)
    sample_rng = prng.split(prng.PRNGKey(seed))[1]
    prompts_all, keys, gids = [], [], []
    for b in range(NB):
        p, _, _ = dp.make_prompts("imagenet_animals", SBS)
        sample_rng, sample_seed = prng.split(sample_rng)
        keys.append(dp.sample_key(prng.split(sample_seed, dp.n_key_devices)))
        prompts_all += p
        gids += [b * world * SBS + rank * SBS + i for i in range(SBS)]         # batch-major global id of my samples
    rewards = np.array([[_reward_of(p, g)] for p, g in zip(prompts_all, gids)])   # (N, 1) float64 like jpeg_fn
    all_r, all_p = dp.gather_rewards(rewards, prompts_all, NB)
    tracker = PerPromptStatTracker(8, 2)
    adv = tracker.update(all_p, all_r)
    adv = dp.local_advantages(adv)
    devs = {k: torch.from_numpy(v) for k, v in _trajectory(gids).items()}
    devs["advantages"] = torch.as_tensor(np.asarray(adv, dtype=np.float32).reshape(-1))
    devs = dp.gather_global(devs, NB)
    outs = []
    for inner in range(2):                                  # two inner epochs: the second shuffles the already shuffled arrays
        devs = dp.shuffle(devs)
        mine = dp.my_rows(devs, TBS)
        outs.append({k: v.numpy().copy() for k, v in mine.items()})
    q.put((rank, prompts_all, np.array(keys), all_r, list(all_p), outs))
    D.barrier()
    dist.destroy_process_group()


def _run(mode):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, mode, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(WORLD)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    return res


def _reference_one_process(n_dev, seed, prompts_fn_name="imagenet_animals"):
    """The reference's statements for one process with n_dev local devices (numpy)."""
    from ddpo_amd.training.prompts import make_prompts           # pinned bit-exact to the reference's prompt functions elsewhere
    from ddpo_amd.utils import prng
    from ddpo_amd.utils.stat_tracking import PerPromptStatTracker
    random.seed(seed)
    np.random.seed(seed)
    sample_rng = prng.split(prng.PRNGKey(seed))[1]
    samples, keys = [], []
    for b in range(NB):
        p, _, _ = make_prompts(prompts_fn_name, n_dev * SBS)                                     # :235-241
        sample_rng, sample_seed = prng.split(sample_rng)                                         # :244
        keys.append(prng.split(sample_seed, n_dev))                                              # :245
        gids = [b * n_dev * SBS + i for i in range(n_dev * SBS)]
        rewards = np.array([[_reward_of(pp, g)] for pp, g in zip(p, gids)])
        samples.append(dict(_trajectory(gids), prompts=np.array(p), rewards=rewards))
    cat = {k: np.concatenate([s[k] for s in samples]) for k in samples[0]}                       # :318-321
    tracker = PerPromptStatTracker(8, 2)
    adv = tracker.update(cat["prompts"], cat["rewards"])                                         # :337
    cat["advantages"] = np.asarray(adv, dtype=np.float32).reshape(1, -1)[0]                      # :349 (process_count = 1)
    prompts, rewards = cat.pop("prompts"), cat.pop("rewards")
    per_inner = []
    for inner in range(2):
        total = cat["log_probs"].shape[0]
        perm = np.random.permutation(total)                                                      # :385
        cat = {k: v[perm] for k, v in cat.items()}
        perms = np.array([np.random.permutation(T) for _ in range(total)])                       # :389-391
        for k in ("latents", "next_latents", "log_probs", "ts"):
            cat[k] = cat[k][np.arange(total)[:, None], perms]
        train = {k: v.reshape(-1, n_dev, TBS, *v.shape[1:]) for k, v in cat.items()}             # :396-399
        per_inner.append([{k: v[:, d].reshape(-1, *v.shape[3:]) for k, v in train.items()} for d in range(n_dev)])
    return prompts, rewards, np.array(keys), per_inner


@pytest.mark.timeout(300)
def test_single_host_world2_equals_one_process_with_two_devices():
    res = _run("single_host")
    prompts, rewards, keys, per_inner = _reference_one_process(WORLD, SEED)
    for rank, my_prompts, my_keys, all_r, all_p, outs in res:
        want_prompts = [p for b in range(NB) for p in prompts[b * WORLD * SBS + rank * SBS: b * WORLD * SBS + (rank + 1) * SBS]]
        assert my_prompts == want_prompts                                    # device d's block of every global prompt batch
        assert np.array_equal(my_keys, keys[:, rank])                        # split(sample_seed, n_devices)[d], bit-exact
        assert np.array_equal(all_r, rewards) and all_p == list(prompts)     # tracker input in the one-process order
        for inner in range(2):
            want = per_inner[inner][rank]
            for k in want:
                assert np.array_equal(outs[inner][k], want[k]), (rank, inner, k)
    # the two ranks together train on every sample exactly once per inner epoch
    ids = np.sort(np.concatenate([r[5][0]["embeds"][:, 0] for r in res]))
    assert np.array_equal(ids, np.arange(WORLD * NB * SBS, dtype=np.float32))


@pytest.mark.timeout(300)
def test_multi_host_world2_keeps_per_process_streams():
    res = _run("multi_host")
    from ddpo_amd.training.prompts import make_prompts
    from ddpo_amd.utils import prng
    all_rewards = []
    for rank, my_prompts, my_keys, all_r, all_p, outs in res:
        random.seed(SEED + rank)                                            # parser.py:177 — seed + process_index
        np.random.seed(SEED + rank)
        sample_rng = prng.split(prng.PRNGKey(SEED + rank))[1]
        want_prompts, want_keys = [], []
        for b in range(NB):
            want_prompts += make_prompts("imagenet_animals", SBS)[0]
            sample_rng, sample_seed = prng.split(sample_rng)
            want_keys.append(prng.split(sample_seed, 1)[0])
        assert my_prompts == want_prompts and np.array_equal(my_keys, np.array(want_keys))
        total = NB * SBS
        perm = np.random.permutation(total)
        assert np.array_equal(outs[0]["embeds"][:, 0] % (WORLD * SBS) // SBS, np.full(total, rank))     # only my own samples
        gids = np.array([b * WORLD * SBS + rank * SBS + i for b in range(NB) for i in range(SBS)], dtype=np.float32)
        assert np.array_equal(outs[0]["embeds"][:, 0], gids[perm])
        all_rewards.append(all_r)
    assert np.array_equal(all_rewards[0], all_rewards[1])                   # process_allgather: rank-major on every rank


def test_mode_validation(monkeypatch):
    from ddpo_amd.training import dp
    monkeypatch.setenv("DDPO_DP_SEMANTICS", "bogus")
    with pytest.raises(ValueError):
        dp.mode_from_env()
    d = dp.DataParallel(mode="single_host", rank=3, world=8)
    assert d.seed_process_index == 0 and d.n_key_devices == 8
    d = dp.DataParallel(mode="multi_host", rank=3, world=8)
    assert d.seed_process_index == 3 and d.n_key_devices == 1


@pytest.mark.parametrize("world,nb,sbs", [(4, 3, 2), (8, 1, 8), (2, 2, 4), (8, 2, 1)])
def test_global_order_turns_rank_major_gather_into_the_one_process_order(world, nb, sbs):
    """all_gather concatenates [rank][batch][sample]; the one-process run holds [batch][device][sample] (each sample batch is unsharded,
    then the batches are concatenated).  Pure index check for larger worlds than the gloo test spawns."""
    from ddpo_amd.training.dp import DataParallel
    gid = lambda b, d, i: (b * world + d) * sbs + i                       # id of sample i of device d in sample batch b, one-process order
    gathered = np.array([gid(b, d, i) for d in range(world) for b in range(nb) for i in range(sbs)])      # what all_gather returns
    order = DataParallel(mode="single_host", rank=0, world=world)._global_order(nb, sbs)
    assert np.array_equal(gathered[order], np.arange(world * nb * sbs))
    # and every device's training rows partition the shuffled global batch
    tb = 1
    total = world * nb * sbs
    rows = [np.arange(total).reshape(-1, world, tb)[:, d].reshape(-1) for d in range(world)]
    assert np.array_equal(np.sort(np.concatenate(rows)), np.arange(total))


@pytest.mark.parametrize("mode,world", [("multi_host", 1), ("multi_host", 2), ("single_host", 2), ("single_host", 4)])
def test_shuffled_rows_equals_shuffle_then_my_rows(mode, world):
    """`shuffled_rows` (one inner epoch: only this rank's rows of the shuffled set are gathered) draws the numpy stream exactly like `shuffle`
    and returns what `my_rows(shuffle(...))` returns, for every rank — so the entrypoint's memory saving changes no sample order."""
    import torch
    from ddpo_amd.training.dp import DataParallel
    total, T, tb = 16, 5, 2
    g = torch.Generator().manual_seed(1)
    base = {"latents": torch.randn(total, T, 4, 2, 2, generator=g), "next_latents": torch.randn(total, T, 4, 2, 2, generator=g),
            "log_probs": torch.randn(total, T, generator=g), "ts": torch.randint(0, 1000, (total, T), generator=g),
            "embeds": torch.randn(total, 7, 3, generator=g), "advantages": torch.randn(total, generator=g)}
    for rank in range(world):
        dp = DataParallel(mode, rank, world)
        r1, r2 = np.random.RandomState(7), np.random.RandomState(7)
        ref = dp.my_rows(DataParallel.shuffle(base, r1), tb)
        got = dp.shuffled_rows(base, tb, r2)
        assert set(got) == set(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
        assert r1.randint(1 << 30) == r2.randint(1 << 30)                      # the stream is left in the same state
        assert got["log_probs"].shape[0] == (total // world if mode == "single_host" else total)
