#!/bin/bash
# Effective shader clock and matrix-pipe occupancy per GEMM / conv kernel instantiation over ONE eager SD-1.5 U-Net forward at batch 16, from
# rocprofv3 counters (round 6): clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration of the SAME pass's kernel trace;
# matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) — the share of the cycles the chip actually ran.  The independent
# check of the timing build's s_memtime / s_memrealtime clocks (profiles/r06_exp1_*.log): the GEMM family runs at 1.2-1.9 GHz, not 2.4.
#   gpurun --timeout 900 -- 'bash tools/pmc_unet_clock.sh'   ->   gpurun_out/pmc_unet_clock.md
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_clock; rm -rf $OUT; mkdir -p $OUT; cd /tmp
pass() { timeout 400 rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o t --output-format csv -- python $R/tools/unet_forward_once.py 1 > $OUT/$1.log 2>&1; }
pass grbm "GRBM_GUI_ACTIVE"
pass sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
cd $OUT
python - <<'PY' | tee $R/gpurun_out/pmc_unet_clock.md
import csv, glob, re, collections
def counters(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    out = collections.defaultdict(dict)
    names = {}
    for row in csv.DictReader(open(f[0])):
        out[row["Dispatch_Id"]][row["Counter_Name"]] = out[row["Dispatch_Id"]].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
    return out, names
def durations(d):
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    out = {}
    for row in csv.DictReader(open(f[0])):
        out[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return out
g, gn = counters("grbm"); gd = durations("grbm")
s, sn = counters("sq"); sd = durations("sq")
def key(name):
    m = re.search(r"(gemm_conv_bf16_buf_kernel|attn_fwd_bf16_dma_kernel|attn_fwd_bf16_pk_kernel)<([^>]*)>", name)
    return None if not m else m.group(1).replace("gemm_conv_bf16_buf_kernel", "gemm") + "<" + m.group(2).replace(" ", "") + ">"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
# the two passes launch the same kernels in the same order: pair them by order of dispatch per kernel key
order_g = collections.defaultdict(list); order_s = collections.defaultdict(list)
for did in sorted(g, key=lambda x: int(x)):
    k = key(gn[did])
    if k: order_g[k].append(did)
for did in sorted(s, key=lambda x: int(x)):
    k = key(sn[did])
    if k: order_s[k].append(did)
print("| kernel instantiation | launches | time in the forward ms | effective clock MHz (GRBM_GUI_ACTIVE / 8 / duration) | matrix pipe busy (of the cycles that ran) | MFMA instructions per launch |")
print("|---|---|---|---|---|---|")
rows = []
for k, dids in order_g.items():
    ga = sum(g[d]["GRBM_GUI_ACTIVE"] for d in dids); ns = sum(gd[d] for d in dids if d in gd)
    sids = order_s.get(k, [])
    mb = sum(s[d].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in sids); mi = sum(s[d].get("SQ_INSTS_MFMA", 0.0) for d in sids)
    ga_s = ga * (len(sids) / max(len(dids), 1))
    rows.append((ns, k, len(dids), ga / 8.0 / ns * 1e3 if ns else float("nan"), mb / (1024.0 * ga_s / 8.0) if ga_s else float("nan"), mi / max(len(sids), 1)))
for ns, k, n, mhz, busy, mi in sorted(rows, reverse=True):
    print(f"| `{k}` | {n} | {ns / 1e6:.2f} | {mhz:.0f} | {100 * busy:.0f} % | {mi:.3g} |")
PY
