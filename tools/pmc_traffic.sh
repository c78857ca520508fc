#!/bin/bash
# HBM traffic of the dominant kernel family via rocprofv3 PMC (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
mkdir -p $OUT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --n-inference-steps 2 --no-cpu-baseline --no-roofline --no-graph ${BENCH_ARGS:-} > $OUT/$c.log 2>&1
done
cd $OUT
python - <<'PY'
import csv, glob, json, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("missing", c); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].split("(")[0]
        fam = "gemm_conv_bf16" if "gemm_conv_bf16" in k else ("gemm_conv_fp32" if "gemm_conv_kernel" in k else ("attention" if "attn" in k else "other"))
        acc[fam][0] += float(row["Counter_Value"])
        key = (fam, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); acc[fam][1] += 1
    res[c] = {k: {"sum_kb": v[0], "launches": v[1]} for k, v in acc.items()}
json.dump(res, open("traffic_raw.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
