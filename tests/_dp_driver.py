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
import torch
from pipeline import policy_gradient as pg

res = pg.main(["--dataset", "compressed-animals", "--resolution", "64", "--n_inference_steps", "4", "--sample_batch_size", "2",
               "--train_batch_size", "2", "--num_train_epochs", "1", "--save_freq", "1", "--per_prompt_stats_min_count", "2",
               "--learning_rate", "1e-4", "--logbase", os.path.join(out, "run")])
from ddpo_amd.utils.serialization import latest_checkpoint
shutil.copy(os.path.join(res["localpath"], f"rewards/{rank}_0.npy"), os.path.join(out, f"rewards_{rank}.npy"))
import gc
objs = [o for o in gc.get_objects() if o.__class__.__name__ == "UNet2DCondition"]
flat = objs[0].params.flat.detach().cpu().numpy()
open(os.path.join(out, f"hash_{rank}.txt"), "w").write(hashlib.sha256(flat.tobytes()).hexdigest())
