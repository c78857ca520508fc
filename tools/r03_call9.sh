#!/bin/bash
# round 3, GPU call: wide 128x320 weight-gradient tile — probe A/B against the 128x128 kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
(cd tools/native && timeout 600 ./kernel_probe wgrad 16 5 > ../../gpurun_out/r03_probe_wgrad.log 2>&1; echo "rc=$?"; cat ../../gpurun_out/r03_probe_wgrad.log | cut -c1-250)
