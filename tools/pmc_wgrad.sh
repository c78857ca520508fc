#!/bin/bash
# SQ-level PMC counters of the bf16x3 weight-gradient kernels (128x128 and wide 128x320 tile) on two layers (separate passes, --kernel-trace only).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_wgrad
mkdir -p $OUT; rm -rf $OUT/*
cd /tmp
python $R/tools/wgrad_once.py
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- python $R/tools/wgrad_once.py > $OUT/g$i.log 2>&1
done
cd $OUT
python - <<'PY'
import csv, glob, os, collections
for d in sorted(glob.glob("g*")):
    if not os.path.isdir(d): continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "gemm_wgrad_bf16" not in k: continue
        tag = k[5:60]
        acc[tag][row["Counter_Name"]] += float(row["Counter_Value"]); n[(tag, row["Counter_Name"])] += 1
    for k, v in acc.items():
        print(d, k, {c: f"{x / max(n[(k, c)], 1):.4g}" for c, x in v.items()})
PY
