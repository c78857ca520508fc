#!/bin/bash
# Per-launch timeline of one sampling step under graph replay (tools/rocpd_timeline.py): which launches of a shape are the slow ones.
#   gpurun --timeout 700 -- 'bash tools/timeline.sh'   ->   gpurun_out/timeline_sampling_step.txt   (~25 s of GPU budget)
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $R/gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train-extra --no-alt-datapath-extra > $R/gpurun_out/prof_tl.log 2>&1
cd $R
python tools/rocpd_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) gpurun_out/timeline_sampling_step.txt
find gpurun_out/prof_tl -name "*.db" -delete
head -3 gpurun_out/timeline_sampling_step.txt
