#!/bin/bash
# One GPU-box round: parity tests (verbose, all failures), smoke, bench, rocprof.  Everything goes to gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=${MAXFAIL:-40} -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
