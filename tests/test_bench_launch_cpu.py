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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["sample", "epoch", "train"])
def test_eight_rank_launch_rehearsal_builds_the_rank_environment_and_the_json_line(mode):
    """the 8-GPU commands the driver will run (`bench.py --gpus 8`, `--mode epoch`, `--mode train`) rehearsed on CPU:
    `--dry-run --backend gloo` self-launches 8 ranks through torch.distributed.run exactly like the real run, builds the process group,
    derives one sampling key per rank from the reference key tree (pairwise distinct), runs barrier / max-over-ranks and the gradient
    all-reduce, and rank 0 prints the mode's JSON line with `rccl_ranks` filled — value null and `dry_run: true`, because no engine
    work runs and nothing is measured."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--mode", mode, "--backend", "gloo", "--dry-run", "--comm-mib", "1",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=_env(), timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["ms_per_step"] is None
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["config"]["parallelism"] == "dp8" and d["scaling"] == "weak"
    assert d["allreduce"]["ranks"] == 8 and d["allreduce"]["sum_correct"] is True
    for k in ("metric", "unit", "steps", "warmup", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in d
    if mode == "sample":
        assert d["metric"] == "sampled images/sec (512^2, 50 DDIM steps)" and d["config"]["global_batch"] == 64
        assert "BASELINE configs[1]" in d["config"]["workload"] and "roofline" in d and "cpu_baseline" in d
    if mode == "epoch":
        assert d["config"]["global_batch"] == 64 and "4 optimizer updates" in d["config"]["workload"]


def test_dry_run_is_refused_without_the_cpu_backend():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
