"""Helper launched by test_gpu_entrypoint.py under torch.distributed.run: runs the entrypoint on LOCAL GPU 0 for every rank."""
import hashlib
import os
import shutil
import sys

import numpy as np

os.environ["LOCAL_RANK_ORIG"] = os.environ.get("LOCAL_RANK", "0")
rank = int(os.environ["RANK"])
os.environ["LOCAL_RANK"] = "0"           # both ranks share the single GPU of the test box
out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import traceback
import torch
try:
    from pipeline import policy_gradient as pg
except Exception:
    open(os.path.join(out, f"error_{rank}.txt"), "w").write(traceback.format_exc())
    raise

def _run():
    return pg.main(["--dataset", "compressed-animals", "--resolution", "64", "--n_inference_steps", "4", "--sample_batch_size", "2",
               "--train_batch_size", "2", "--num_train_epochs", "1", "--save_freq", "1", "--per_prompt_stats_min_count", "2",
               "--learning_rate", "1e-4", "--logbase", os.path.join(out, "run")])


try:
    res = _run()
except BaseException:
    open(os.path.join(out, f"error_{rank}.txt"), "w").write(traceback.format_exc())
    raise
from ddpo_amd.utils.serialization import latest_checkpoint
shutil.copy(os.path.join(res["localpath"], f"rewards/{rank}_0.npy"), os.path.join(out, f"rewards_{rank}.npy"))
shutil.copy(os.path.join(res["localpath"], f"prompts/{rank}_0.npy"), os.path.join(out, f"prompts_{rank}.npy"))
flat = res["state"].params.flat.detach().cpu().numpy()
info = np.load(os.path.join(res["localpath"], "train_info/0_0_0.npy"), allow_pickle=True).item() if rank == 0 else None
if rank == 0:
    assert float(np.max(info["approx_kl"])) < 1e-8, info      # unchanged weights: the stored log-probs are reproduced
open(os.path.join(out, f"hash_{rank}.txt"), "w").write(hashlib.sha256(flat.tobytes()).hexdigest())
