#!/usr/bin/env python
"""Per-kernel register / spill / LDS summary of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
usage: python tools/kernel_resources.py ddpo_amd/csrc/gemm_bf16.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'vspill':>6} {'SGPR':>5} {'sspill':>6} {'occ':>3} {'scratch':>7}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    if flt and flt not in n:
        continue
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} {r.get('TotalSGPRs', '?'):>5} {r.get('SGPRs Spill', '?'):>6} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>3} {r.get('ScratchSize [bytes/lane]', '?'):>7}  {n}")
