#!/bin/bash
# SQ-level PMC passes only (issue / LDS / MFMA occupancy) for the probe shapes; prints per-launch means.
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc/*
cd /tmp
run() { # name, counters, kind
  timeout 300 rocprofv3 --kernel-trace --pmc $2 -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$1 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/probe_gemm.py $3 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$1.log 2>&1
}
for kind in ${KINDS:-conv gemm}; do
  run ${kind}_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" $kind
  run ${kind}_sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $kind
  run ${kind}_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" $kind
done
cd $GRAFT_REPO_ROOT/gpurun_out/pmc
python - <<'PY'
import csv, glob, os, collections
for d in sorted(glob.glob("*_*")):
    if not os.path.isdir(d): continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(d, "no counter file; log tail:"); os.system(f"tail -3 {d}.log"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"][:48]
        if "gemm" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        print(d, k, {c: f"{x/4:.4g}" for c, x in v.items()})
PY
