#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_exp5.log
{
echo "== pytest (staging kernel, copy_cols in the captured step, cluster PPO)"
DDPO_HEADLINE_STEPS=2 DDPO_TRAJ_STEPS=3 timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_headline_geometry.py tests/test_gpu_entrypoint.py tests/test_fused_micro_steps.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6

} > $L 2>&1
tail -12 $L
