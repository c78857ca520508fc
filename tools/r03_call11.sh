#!/bin/bash
# round 3, GPU call: VAE decoder convolutions in the probe (k-blocked weights as the model uses them)
mkdir -p gpurun_out; export TMPDIR=/tmp
cd tools/native && PROBE_WKBLK=1 timeout 600 ./kernel_probe vae 8 3 > ../../gpurun_out/r03_probe_vae.log 2>&1; echo "rc=$?"; cut -c1-230 ../../gpurun_out/r03_probe_vae.log
