#!/bin/bash
# Second closing run of round 4 (after the output-stage rewrite and ABI v12): the GPU tests not yet re-run on the new library (everything
# except the files run by tools/r04_ab_epilogue.sh and the four slowest full-size tests), one full-size train parity test, smoke.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
DDPO_PARITY_LOG=$R/gpurun_out/r04_parity_margins_close2.log timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 \
  --ignore=tests/test_gpu_planes.py --ignore=tests/test_gpu_bf16.py --ignore=tests/test_gpu_kernels.py --ignore=tests/test_gpu_model.py \
  -k "not headline and not sd21_full_size and not rwr_step_sd15_full_size" > gpurun_out/r04_pytest_close2.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_pytest_close2.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" gpurun_out/r04_pytest_close2.log | cut -c1-220 | tail -12
cat gpurun_out/r04_parity_margins_close2.log 2>/dev/null | cut -c1-260
timeout 300 python __graft_entry__.py smoke > gpurun_out/r04_smoke_close2.log 2>&1; echo "smoke exit $?" >> gpurun_out/r04_smoke_close2.log; tail -2 gpurun_out/r04_smoke_close2.log
