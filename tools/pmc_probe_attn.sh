#!/bin/bash
# SQ-level PMC counters of the attention forward kernels (register-staged packed-image kernel vs LDS-DMA kernel) on the torch-free probe.
#   gpurun --timeout 300 -- 'bash tools/pmc_probe_attn.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_probe_attn
mkdir -p $OUT; rm -rf $OUT/*
cd /tmp
for dm in 0 1; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    DDPO_ATTN_DMA=$dm timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/dma${dm}_g$i -o p --output-format csv -- \
      $R/tools/native/kernel_probe attn 16 3 > $OUT/dma${dm}_g$i.log 2>&1
  done
done
cd $OUT
python - <<'PY'
import csv, glob, os, collections
for d in sorted(glob.glob("dma*_g*")):
    if not os.path.isdir(d): continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "attn_fwd_bf16_pk_kernel<40" not in k and "attn_fwd_bf16_dma_kernel<40" not in k: continue
        if int(row.get("Grid_Size", "0") or 0) and int(row["Grid_Size"]) < 1000000: continue      # keep the 4096 x 4096 self-attention launches only
        tag = k[5:45]
        acc[tag][row["Counter_Name"]] += float(row["Counter_Value"]); n[(tag, row["Counter_Name"])] += 1
    for k, v in acc.items():
        print(d, k, {c: f"{x / max(n[(k, c)], 1):.4g}" for c, x in v.items()})
PY
