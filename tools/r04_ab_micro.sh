#!/bin/bash
# LayerNorm rows kernel with interleaved reductions + vectorised split-K reduce: kernel / plane / model tests, then the sampling bench against
# the previous library (tools/native/libddpo_hip_v3.so), interleaved on one box.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_planes.py tests/test_gpu_f16mx_model.py tests/test_gpu_bf16.py -m gpu -q -x -p no:cacheprovider -k "not attention and not wgrad" 2>&1 | tail -3
L=ddpo_amd/libddpo_hip.so
cp $L /tmp/new.so
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for round in 1 2; do
  for v in v3 new; do
    if [ $v = new ]; then cp /tmp/new.so $L; else cp tools/native/libddpo_hip_$v.so $L; fi
    line=$(timeout 400 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "sample lib=$v (round $round): $line" | tee -a gpurun_out/r04_ab_micro.log
  done
done
cp /tmp/new.so $L
