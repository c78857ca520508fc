#!/bin/bash
# HBM-side traffic of the GEMM / conv kernel family over EXACTLY the launches of N U-Net forwards (no VAE, no sampler glue):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (--pmc with --kernel-trace only), gfx950 correction FETCH x2
# (MI355X_MICROARCH.md "HBM").  Writes gpurun_out/traffic_unet/traffic_unet.json.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_unet
mkdir -p $OUT; cd /tmp
N=${N:-2}
python $GRAFT_REPO_ROOT/tools/unet_forward_once.py $N > $OUT/algorithmic.json 2> $OUT/algorithmic.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/unet_forward_once.py $N > $OUT/$c.log 2>&1
done
cd $OUT
python - <<'PY'
import csv, glob, json, collections
alg = json.loads(open("algorithmic.json").read().strip().splitlines()[-1])
res = {"algorithmic": alg}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("missing", c); continue
    tot, seen = 0.0, set()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "gemm_conv" not in k:
            continue
        tot += float(row["Counter_Value"]); seen.add(row["Dispatch_Id"])
    res[c] = {"sum_kb": tot, "launches": len(seen)}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    n = res["FETCH_SIZE"]["launches"]
    res["fetch_bytes_per_launch"] = 2.0 * 1024.0 * res["FETCH_SIZE"]["sum_kb"] / n          # gfx950: x2 for 16-B/lane streaming reads
    res["write_bytes_per_launch"] = 1024.0 * res["WRITE_SIZE"]["sum_kb"] / res["WRITE_SIZE"]["launches"]
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / alg["algorithmic_bytes_per_launch"]
# stamp: what bench.py checks before it quotes this figure (content hashes of the kernel sources the counters were taken on)
import hashlib, os, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cs = os.path.join(root, "ddpo_amd", "csrc")
res["csrc_sha256"] = {f: hashlib.sha256(open(os.path.join(cs, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(cs)) if f.endswith((".hip", ".h"))}
res["collected_unix"] = int(time.time())
res["collected_date"] = time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime())
json.dump(res, open("traffic_unet.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
