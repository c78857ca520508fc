#!/bin/bash
# SQ-level PMC counters of the fp32-fed vs the plane-fed GEMM kernel on ONE layer, collected on the torch-free probe (a PMC
# pass costs seconds instead of a Python start-up).  Separate rocprofv3 passes per counter group; never combined with trace domains
# other than --kernel-trace.
#   gpurun --timeout 300 -- 'CASE=c0 MODES="6 7" bash tools/pmc_probe.sh'        (c0 = conv 3x3 320->320 @ 64^2, d1 = dense 65536x320x2560 ...)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_probe
mkdir -p $OUT; rm -rf $OUT/*
cd /tmp
for m in ${MODES:-6}; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    PROBE_ONLY=${CASE:-c0} DDPO_APL_MODE=$m timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/mode${m}_g$i -o p --output-format csv -- \
      $R/tools/native/kernel_probe gemm2 16 4 > $OUT/mode${m}_g$i.log 2>&1
  done
done
cd $OUT
python - <<'PY'
import csv, glob, os, collections
for d in sorted(glob.glob("mode*_g*")):
    if not os.path.isdir(d): continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(d, "no counter file; log tail:"); os.system(f"tail -3 {d}.log"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "gemm_conv_bf16_buf_kernel" not in k: continue
        tag = "planes " if "true, " in k[-12:] or k.rstrip(">").split(",")[-1].strip() not in ("0", "false") else "fp32-fed"
        acc[tag + k[30:62]][row["Counter_Name"]] += float(row["Counter_Value"]); n[(tag + k[30:62], row["Counter_Name"])] += 1
    for k, v in acc.items():
        print(d, k, {c: f"{x / max(n[(k, c)], 1):.4g}" for c, x in v.items()})
PY
