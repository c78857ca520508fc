#!/usr/bin/env python
"""Pick the hyper-parameters of tests/test_gpu_learning.py on hardware: a few (lr, batch) settings of the tiny jpeg run, reward
curve summary per setting (first / last 5 epochs, z-score, seconds)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_learning import run_learning, _gain
grid = [("compressed-animals", "1e-4", 16, 8), ("compressed-animals", "3e-4", 16, 8), ("compressed-animals", "1e-3", 16, 8),
        ("compressed-animals", "3e-4", 16, 16), ("neg-compressed-animals", "3e-4", 16, 8), ("neg-compressed-animals", "1e-3", 16, 8)]
if len(sys.argv) > 1:
    grid = [tuple(a.split(",")) for a in sys.argv[1:]]
epochs = int(os.environ.get("DDPO_LEARN_EPOCHS", "40"))
print("| dataset | lr | sample bs | train bs | first 5 | last 5 | gain | z | s |\n|---|---|---|---|---|---|---|---|---|")
for ds, lr, sbs, tbs in grid:
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        so = sys.stdout; sys.stdout = open(os.devnull, "w")
        try:
            r, _ = run_learning(d, ds, epochs=epochs, lr=lr, sbs=sbs, tbs=tbs)
        finally:
            sys.stdout = so
    g, z = _gain(r)
    print(f"| {ds} | {lr} | {sbs} | {tbs} | {r[:5].mean():.4f} | {r[-5:].mean():.4f} | {g:+.4f} | {z:.1f} | {time.time() - t0:.0f} |", flush=True)
