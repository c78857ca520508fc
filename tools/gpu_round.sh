#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats.  Everything goes to gpurun_out/.
#   SKIP_TESTS=1 / SKIP_BENCH=1 / SKIP_PROF=1 skip a stage; PROBE=1 prepends the torch-free kernel probe (seconds);
#   AB=1 appends the in-model fp32-fed vs plane-fed per-layer comparison and a DDPO_PLANES=1 bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${PROBE:-0}" = "1" ]; then bash tools/probe_round.sh; fi
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1500 python -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-40} -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log; tail -5 gpurun_out/bench.log
fi
if [ "${AB:-0}" = "1" ]; then
timeout 300 python tools/unet_gemm_breakdown.py 16 --ab > gpurun_out/gemm_breakdown_ab.log 2>&1; tail -40 gpurun_out/gemm_breakdown_ab.log
DDPO_PLANES=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_planes.log 2>&1; tail -1 gpurun_out/bench_planes.log | cut -c1-300
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/prof.log; find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
# keep only the small summaries (the full trace can be large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
