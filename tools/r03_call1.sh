#!/bin/bash
# round 3, GPU call 1: k-loop variants in the torch-free probe (interleaved A/B, warm and cold), the new C5 / bfloat16 tests, baseline bench
mkdir -p gpurun_out; export TMPDIR=/tmp
P=tools/native/kernel_probe
{
for cold in 0 1; do for rep in 1 2; do
  echo "== rep=$rep cold=$cold TALL_ROT=0 APL_MODE=7"; PROBE_COLD=$cold DDPO_TALL_ROT=0 DDPO_APL_MODE=7 timeout 120 $P gemm2 16 10 | grep -v "^#"
  echo "== rep=$rep cold=$cold TALL_ROT=1 APL_MODE=7"; PROBE_COLD=$cold DDPO_TALL_ROT=1 DDPO_APL_MODE=7 timeout 120 $P gemm2 16 10 | grep -v "^#"
  echo "== rep=$rep cold=$cold TALL_ROT=1 APL_MODE=15"; PROBE_COLD=$cold DDPO_TALL_ROT=1 DDPO_APL_MODE=15 timeout 120 $P gemm2 16 10 | grep -v "^#"
done; done
} > gpurun_out/r03_probe_kloop.log 2>&1
grep -c "bit-identical" gpurun_out/r03_probe_kloop.log; grep -c FAIL gpurun_out/r03_probe_kloop.log
timeout 1500 python -m pytest tests/test_gpu_train_parity.py::test_train_step_sd21_full_size_bf16x3 tests/test_gpu_model.py::test_sampler_sd21_full_size_96x96_graph_path tests/test_gpu_bf16.py::test_bfloat16_dtype_path_is_held_to_the_reference_bf16_arithmetic tests/test_gpu_planes.py -m gpu -q -s -p no:cacheprovider -x > gpurun_out/r03_pytest_c5.log 2>&1; grep -E "train parity|sd21 96x96|bfloat16 dtype|passed|failed|Error" gpurun_out/r03_pytest_c5.log | cut -c1-400
for v in "0 7" "1 7" "1 15" "0 7" "1 7" "1 15"; do set -- $v
  DDPO_TALL_ROT=$1 DDPO_APL_MODE=$2 timeout 300 python bench.py --no-cpu-baseline --no-train-extra --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample TALL_ROT=$1 APL_MODE=$2', d['value'], d['ms_per_step'])"
done | tee gpurun_out/r03_ab_kloop_bench.log
