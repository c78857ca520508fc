#!/bin/bash
# Per GEMM / conv kernel instantiation over ONE eager SD-1.5 U-Net forward at batch 16 (tools/unet_forward_once.py): the XCD L2's hit rate on the
# operand stream (TCC_HIT / TCC_MISS) and how busy the matrix pipe is (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), the
# normalisation of profiles/r02_pmc_sq_gemm.md) — the two numbers behind
# "these kernels wait for the L2 -> LDS stream, not for MFMAs" (DESIGN section 6.0).  Separate --pmc passes with --kernel-trace only.
#   gpurun --timeout 900 -- 'bash tools/pmc_unet_l2_mfma.sh'   ->   gpurun_out/pmc_unet_l2_mfma.md   (~2 min of budget)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_l2; rm -rf $OUT; mkdir -p $OUT; cd /tmp
pass() { timeout 400 rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o t --output-format csv -- python $R/tools/unet_forward_once.py 1 > $OUT/$1.log 2>&1; }
pass l2 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
pass sq "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
pass grbm "GRBM_GUI_ACTIVE"
cd $OUT
python - <<'PY' | tee $R/gpurun_out/pmc_unet_l2_mfma.md
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for d in ("l2", "sq", "grbm"):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("missing pass", d); continue
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        m = re.search(r"gemm_conv_bf16_buf_kernel<([^>]*)>", k)
        if not m: continue
        key = m.group(1).replace(" ", "")
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
        if (d, row["Dispatch_Id"]) not in seen and d == "l2":
            seen.add((d, row["Dispatch_Id"])); n[key] += 1
print("| instantiation <BM,BN,NPASS,ABL,WM,WN,DEEP,APL> | launches | L2 hit rate | memory-side 64 B read requests per launch | matrix pipe busy | waves issuing / stalled on issue / parked (of wave cycles) |")
print("|---|---|---|---|---|---|")
for key, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    g = lambda a, b: f"{v.get(a, 0) / v[b]:.2f}" if v.get(b) else "-"
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0) if v.get("GRBM_GUI_ACTIVE") else float("nan")
    print(f"| `{key}` | {n[key]} | {hit / max(hit + miss, 1):.2f} | {v.get('TCC_EA0_RDREQ_sum', 0) / max(n[key], 1):.3g} | {100 * busy:.0f} % | {g('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES')} / {g('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES')} / {g('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')} |")
PY
