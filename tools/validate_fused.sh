#!/bin/bash
# Hardware validation of the fused PPO micro-steps (DDPO_TRAIN_FUSE / train_steps_fused); results go to gpurun_out/.
#   gpurun --timeout 900 -- 'bash tools/validate_fused.sh'
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_micro_steps.py tests/test_gpu_kernels.py -k "fused or grouped or ppo" \
  -m gpu -q -p no:cacheprovider > gpurun_out/fused_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/fused_tests.log; tail -5 gpurun_out/fused_tests.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
for k in 1 4 8; do
  timeout 600 python bench.py --mode train --train-fuse $k --steps 8 --warmup 4 > gpurun_out/bench_train_fuse$k.log 2>&1
  tail -1 gpurun_out/bench_train_fuse$k.log | cut -c1-400
done
fi
