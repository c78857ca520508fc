mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest "tests/test_gpu_train_parity.py::test_train_step_at_the_reference_clip_range" tests/test_gpu_entrypoint.py tests/test_gpu_rccl_single_rank.py -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest_fixups.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_pytest_fixups.log; tail -5 gpurun_out/r04_pytest_fixups.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/r04_bench_final2.log 2>&1; echo "exit $?" >> gpurun_out/r04_bench_final2.log; tail -2 gpurun_out/r04_bench_final2.log | cut -c1-600
