#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_backward.py tests/test_fused_micro_steps.py tests/test_gpu_train_parity.py -m gpu -q -p no:cacheprovider > gpurun_out/r02_pytest_train_planes.log 2>&1; tail -8 gpurun_out/r02_pytest_train_planes.log | cut -c1-300
for rep in 1 2; do for po in 0 1; do
  DDPO_PLANES_OUT=$po timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample PLANES_OUT=$po', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r02_ab_planes_out.log
for rep in 1 2; do for tp in 0 1; do
  DDPO_TRAIN_PLANES=$tp timeout 300 python bench.py --mode train --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train TRAIN_PLANES=$tp', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r02_ab_train_planes.log
timeout 300 python tools/unet_gemm_breakdown.py 16 --ab > gpurun_out/r02_gemm_breakdown_ab2.log 2>&1; head -4 gpurun_out/r02_gemm_breakdown_ab2.log
DDPO_LEARN_EPOCHS=100 DDPO_ALLOW_SYNTHETIC=1 timeout 400 python tools/learning_sweep.py compressed-animals,3e-4,16,8 neg-compressed-animals,3e-4,16,8 compressed-animals,3e-4,32,16 neg-compressed-animals,3e-4,32,16 neg-compressed-animals,3e-4,16,16 > gpurun_out/r02_learning_sweep2.md 2>gpurun_out/r02_learning_sweep2.err; cat gpurun_out/r02_learning_sweep2.md; tail -2 gpurun_out/r02_learning_sweep2.err
