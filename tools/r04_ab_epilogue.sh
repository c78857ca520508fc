#!/bin/bash
# Output stage of the buffer-addressed GEMM kernels (EpiRows, csrc/gemm_bf16.hip): bit-identity + timing through the probe, the GEMM / plane /
# model tests, then the sampling bench against private libraries built from earlier gemm_bf16.hip revisions (tools/native/libddpo_hip_<tag>.so:
# prev = per-iteration loads, v1 = batched loads in one phase), interleaved on one box.  HANDOVER="0 1" also toggles the plane hand-over.
mkdir -p gpurun_out; export TMPDIR=/tmp
(cd tools/native && PROBE_WKBLK=1 timeout 200 ./kernel_probe gemm2 16 20 | cut -c1-170) 2>&1 | tee gpurun_out/r04_probe_gemm2_epilogue.log | tail -28
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_bf16.py tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "not full_size and not sd15 and not sd21" 2>&1 | tail -4
L=ddpo_amd/libddpo_hip.so
cp $L /tmp/new.so
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
for round in 1 2; do
  for v in ${TAGS:-prev v1} new; do
    if [ $v = new ]; then cp /tmp/new.so $L; else cp tools/native/libddpo_hip_$v.so $L; fi
    for h in ${HANDOVER:-0}; do
      line=$(DDPO_ATTN_PLANES=$h DDPO_H3_PLANES=$h timeout 400 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
      echo "sample gemm=$v handover=$h (round $round): $line" | tee -a gpurun_out/r04_ab_epilogue.log
    done
  done
done
cp /tmp/new.so $L
