#!/bin/bash
# First GPU call of round 5: the four experiments prepared at the end of round 4 (RUNBOOK "Next" 0), each through its probe / parity tests and
# all of them through ONE interleaved bench A/B.  Build the private libraries locally first (cross-compiles, ~2 min):
#     bash tools/r05_first_call.sh prepare
#     gpurun --timeout 1500 -- 'bash tools/r05_first_call.sh'            (~6 min of budget)
cd "$(dirname "$0")/.." || exit 1
if [ "$1" = prepare ]; then
  set -e
  make -C ddpo_amd/csrc && make -C tools/native kernel_probe
  bash tools/native/build_variant_lib.sh fgelu -DDDPO_EXP_FAST_GELU
  bash tools/native/build_variant_lib.sh episgpr -DDDPO_EXP_EPI_SGPR
  bash tools/native/build_variant_lib.sh dpp -DDDPO_EXP_DPP_REDUCE
  bash tools/native/build_variant_lib.sh attnlds -DDDPO_EXP_ATTN_PLANE_LDS
  exit 0
fi
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/r05_first_call.log; : > $LOG
(cd tools/native && timeout 60 ./kernel_probe gelu; echo "gelu probe exit $?"; timeout 60 ./kernel_probe reduce | grep -v "lanes 0..63 read"; echo "reduce probe exit ${PIPESTATUS[0]}") 2>&1 | tee -a $LOG
L=ddpo_amd/libddpo_hip.so; cp $L /tmp/new.so
run_tests() {        # <tag> <pytest args>: the parity tests on a private build
  cp tools/native/libddpo_hip_$1.so $L
  echo "== pytest on lib=$1: $2" | tee -a $LOG
  eval "timeout 900 python -m pytest $2 -m gpu -q -x -p no:cacheprovider" 2>&1 | tail -2 | tee -a $LOG
  cp /tmp/new.so $L
}
run_tests fgelu "tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_backward.py -k 'not full_size and not sd15 and not sd21'"
run_tests episgpr "tests/test_gpu_planes.py tests/test_gpu_kernels.py -k 'gemm or conv or linear or planes'"
run_tests attnlds "tests/test_gpu_bf16.py tests/test_gpu_model.py -k 'plane_emitting or plane_handover'"
TAGS="fgelu episgpr attnlds"
grep -q "reduce probe exit 0" $LOG && { run_tests dpp "tests/test_gpu_kernels.py tests/test_gpu_planes.py -k 'norm or ddim or ppo'"; TAGS="$TAGS dpp"; }
TAGS="$TAGS" ROUNDS=2 LOG=r05_first_call_ab.log bash tools/ab_bench.sh 2>&1 | tail -12 | tee -a $LOG
