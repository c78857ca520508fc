#!/usr/bin/env python
"""Merge a PMC traffic run (tools/pmc_unet_traffic.sh -> gpurun_out/traffic_unet/traffic_unet.json) into profiles/roofline_traffic.json as the
entry bench.py quotes (`<datapath>_unet`), stamped with the git commit it was taken on.  bench.py refuses the entry when the kernel sources
have changed since (content hashes recorded by the collection script on the GPU box).   usage: python tools/stamp_traffic.py [datapath]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = sys.argv[1] if len(sys.argv) > 1 else "f16mx"          # the shipped datapath (ddpo_amd.lib.SHIPPED_DATAPATH)
src = os.path.join(ROOT, "gpurun_out", "traffic_unet", "traffic_unet.json")
raw = json.load(open(src))
if "traffic_bytes_per_launch" not in raw:
    raise SystemExit(f"{src}: incomplete run (no traffic_bytes_per_launch)")
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "ddpo_amd/csrc"], capture_output=True, text=True).stdout.strip())
dst = os.path.join(ROOT, "profiles", "roofline_traffic.json")
allj = json.load(open(dst)) if os.path.exists(dst) else {}
allj[dp + "_unet"] = {
    "kernel": "gemm_conv_bf16_* family (plane-fed + fp32-fed, all tile shapes): ONLY the launches of the eager SD-1.5 U-Net forwards at batch 16 "
              "(tools/unet_forward_once.py) — the launches bench.py's roofline pass event-times; no VAE, no sampler glue",
    "launches": raw["FETCH_SIZE"]["launches"],
    "fetch_bytes_per_launch": raw["fetch_bytes_per_launch"], "write_bytes_per_launch": raw["write_bytes_per_launch"],
    "traffic_bytes_per_launch": raw["traffic_bytes_per_launch"],
    "algorithmic_bytes_per_launch": raw["algorithmic"]["algorithmic_bytes_per_launch"], "traffic_over_algorithmic": raw["traffic_over_algorithmic"],
    "note": "PMC: FETCH_SIZE x2 (gfx950 16-B/lane correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE, separate rocprofv3 --pmc passes with --kernel-trace only "
            "(tools/pmc_unet_traffic.sh); memory-side L2 requests, Infinity-Cache hits included",
    "collected_unix": raw["collected_unix"], "collected_date": raw["collected_date"], "csrc_sha256": raw["csrc_sha256"],
    "git_commit": commit + ("+dirty-csrc" if dirty else ""),
}
json.dump(allj, open(dst, "w"), indent=1)
print(json.dumps(allj[dp + "_unet"], indent=1))
