"""Launched by tests/test_gpu_rccl_single_rank.py under torch.distributed.run with ONE rank and DDPO_FORCE_DIST=1: every collective
of the data-parallel path goes through a real RCCL communicator (backend "nccl") on the box's GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

from ddpo_amd.training import distributed as D
from ddpo_amd.training.dp import DataParallel

rank, world = D.init()
assert dist.is_initialized() and dist.get_backend() == "nccl" and (rank, world) == (0, 1)
dev = torch.device("cuda", 0)
out = {}
r = D.allgather_array(np.arange(6, dtype=np.float64).reshape(3, 2))
out["allgather_array"] = r.tolist()
out["allgather_strings"] = D.allgather_strings(["a cat", "a dog"])
t = torch.arange(12, dtype=torch.float32, device=dev).reshape(3, 4)
g = D.allgather_tensor(t)
out["allgather_tensor_ok"] = bool(torch.equal(g, t)) and g.device.type == "cuda"
flat = torch.ones(16 << 20, dtype=torch.float32, device=dev)              # 64 MiB through RCCL's all-reduce
D.allreduce_sum_(flat)
torch.cuda.synchronize()
out["allreduce_ok"] = bool((flat == 1.0).all().item())
out["pmean"] = D.pmean_info({"loss": torch.tensor(2.0, device=dev), "kl": 0.5})
dp = DataParallel(mode="single_host")
devs = {"log_probs": torch.randn(4, 3, device=dev), "latents": torch.randn(4, 3, 2, device=dev), "next_latents": torch.randn(4, 3, 2, device=dev),
        "ts": torch.zeros(4, 3, dtype=torch.int64, device=dev), "embeds": torch.randn(4, 5, device=dev), "advantages": torch.randn(4, device=dev)}
gl = dp.gather_global(devs)
out["gather_global_ok"] = all(torch.equal(gl[k], devs[k]) for k in devs)
D.barrier()
out["backend"] = dist.get_backend()
out["nccl_version"] = list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None
print("RCCL_SMOKE " + json.dumps(out), flush=True)
dist.destroy_process_group()
