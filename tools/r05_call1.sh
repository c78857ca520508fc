#!/bin/bash
# round 5, call 1: the four prepared experiments (tools/r05_first_call.sh) + the long-trajectory parity tests (all 50 steps at full size, once)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/r05_first_call.sh
export DDPO_PARITY_LOG=$PWD/gpurun_out/r05_parity_trajectory.log; : > $DDPO_PARITY_LOG
DDPO_TRAJ_STEPS=50 timeout 1200 python -m pytest tests/test_gpu_headline_geometry.py::test_headline_size_trajectory_error_growth_over_many_steps "tests/test_gpu_model.py::test_sampler_50_steps_matches_oracle" -m gpu -q -x -p no:cacheprovider --durations=5 2>&1 | tail -15 | tee gpurun_out/r05_pytest_trajectory.log
