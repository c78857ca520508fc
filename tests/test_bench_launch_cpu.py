"""bench.py's launch contract, on CPU: `--gpus N` starts its own N ranks (torch.distributed.run) when it is not already under a
launcher, reports what the process group saw (`n_gpus`, `rccl_ranks`) and times the gradient all-reduce; it refuses to run fewer
ranks than asked.  The collective-only mode (`--mode comm`) is the part that can execute without the HIP engine, on gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_gpus_n_self_launches_n_ranks_and_times_the_allreduce(world):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--mode", "comm", "--backend", "gloo", "--comm-mib", "1", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, env=_env(), timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout              # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["rccl_ranks"] == world and d["config"]["parallelism"] == f"dp{world}"
    ar = d["allreduce"]
    assert ar["ranks"] == world and ar["sum_correct"] is True and ar["bytes"] == 1 << 20 and ar["ms"] > 0
    assert ar["busbw_GBps"] == pytest.approx(ar["algbw_GBps"] * 2 * (world - 1) / world)          # ring bus bandwidth: 2 (W-1)/W of the algorithm bandwidth
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d


def test_more_gpus_than_the_node_has_is_refused_loudly():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    ask = max(have, 1) + 1                       # one more than the node has (and at least 2: --gpus 1 never self-launches)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(ask)], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout) and f"--gpus {ask}" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no result line of any kind


def test_world_size_must_match_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--mode", "comm", "--backend", "gloo"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
