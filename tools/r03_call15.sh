#!/bin/bash
# round 3, GPU call: opt-in f16mx datapath at model level + regression of the bf16x3 suites after the lib.py routing changes
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_f16mx_model.py tests/test_gpu_f16mx.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r03_pytest_f16mx_model.log 2>&1; tail -30 gpurun_out/r03_pytest_f16mx_model.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_fused_micro_steps.py tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_model.py::test_sampler_sd21_full_size_96x96_graph_path > gpurun_out/r03_pytest_regress.log 2>&1; tail -4 gpurun_out/r03_pytest_regress.log | cut -c1-250
