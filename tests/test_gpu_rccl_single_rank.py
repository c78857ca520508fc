"""The RCCL path executed on hardware.  A 1-GPU box cannot host two RCCL ranks (one rank per device), so the multi-rank logic is
covered by the gloo tests (world 2, CPU) and by the 2-rank entrypoint test that shares the GPU over gloo; what THIS test adds is that
the `nccl` backend — RCCL on ROCm — really initialises, binds its communicator to the device and carries every collective of the
data-parallel path (all_reduce of the gradient buffer, all_gather of rewards / prompts / trajectories, barrier, the benchmark's
max-over-ranks) with one rank, through the same code the N-rank job runs (DDPO_FORCE_DIST=1 builds the group for world size 1) —
including, since round 4, the DEFAULT gradient path of world > 1: the bucketed all-reduce on a side stream behind the backward pass
(training/distributed.GradBucketer), against the blocking single all-reduce."""
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
    # the default data-parallel gradient path (GradBucketer: side stream + async all_reduce per bucket + finish) executed on RCCL:
    # same applied update as the blocking all-reduce bit for bit on fixed gradients, several buckets launched BEFORE finish();
    # and through train_step / train_steps_fused (UNet.backward(on_ready=...)) equal up to the order of the weight gradients' fp32 atomics
    assert out["bucketed_update_bit_equal"] and out["buckets"] > 8 and 0 < out["launched_before_finish"] < out["buckets"]
    assert out["train_step_update_l2"] > 1e-3 and out["train_step_overlap_l2_diff"] < 1e-2 * out["train_step_update_l2"], out
    assert abs(out["train_step_loss"][0] - out["train_step_loss"][1]) < 1e-4 * abs(out["train_step_loss"][0]) + 1e-6


@pytest.mark.timeout(600)
def test_bench_comm_mode_reports_the_rccl_world():
    p = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "comm", "--comm-mib", "256", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rccl_ranks"] == 1 and d["allreduce"]["backend"] == "nccl" and d["allreduce"]["sum_correct"]
    assert d["allreduce"]["bytes"] == 256 * (1 << 20) and d["allreduce"]["ms"] > 0
