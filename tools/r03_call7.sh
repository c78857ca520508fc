#!/bin/bash
# round 3, GPU call: f16mx plane-fed datapath — probe: contract check against the decoded planes, accuracy, timing vs bf16x3
mkdir -p gpurun_out; export TMPDIR=/tmp
cd tools/native
timeout 600 ./kernel_probe mx 16 10 > ../../gpurun_out/r03_probe_mx.log 2>&1
echo "rc=$?"
tail -30 ../../gpurun_out/r03_probe_mx.log | cut -c1-260
