"""The RCCL path executed on hardware.  A 1-GPU box cannot host two RCCL ranks (one rank per device), so the multi-rank logic is
covered by the gloo tests (world 2, CPU) and by the 2-rank entrypoint test that shares the GPU over gloo; what THIS test adds is that
the `nccl` backend — RCCL on ROCm — really initialises, binds its communicator to the device and carries every collective of the
data-parallel path (all_reduce of the gradient buffer, all_gather of rewards / prompts / trajectories, barrier, the benchmark's
max-over-ranks) with one rank, through the same code the N-rank job runs (DDPO_FORCE_DIST=1 builds the group for world size 1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, port=None):
    port = port or _free_port()
    env = dict(os.environ, DDPO_FORCE_DIST="1", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("DDPO_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=540)


@pytest.mark.timeout(600)
def test_collectives_through_rccl_with_one_rank():
    p = _launch([os.path.join(ROOT, "tests", "_rccl_driver.py")])
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_SMOKE ")][-1]
    out = json.loads(line[len("RCCL_SMOKE "):])
    assert out["backend"] == "nccl"
    assert out["allgather_array"] == [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]] and out["allgather_strings"] == ["a cat", "a dog"]
    assert out["allgather_tensor_ok"] and out["allreduce_ok"] and out["gather_global_ok"]
    assert abs(out["pmean"]["loss"] - 2.0) < 1e-6 and abs(out["pmean"]["kl"] - 0.5) < 1e-6


@pytest.mark.timeout(600)
def test_bench_comm_mode_reports_the_rccl_world():
    p = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "comm", "--comm-mib", "256", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rccl_ranks"] == 1 and d["allreduce"]["backend"] == "nccl" and d["allreduce"]["sum_correct"]
    assert d["allreduce"]["bytes"] == 256 * (1 << 20) and d["allreduce"]["ms"] > 0
