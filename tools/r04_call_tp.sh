mkdir -p gpurun_out/r04c
timeout 900 python tools/r04_ab_two_pass_own.py 2>&1 | tee gpurun_out/r04c/ab_two_pass_own.txt
timeout 900 python -m pytest tests/test_two_pass_gpu.py tests/test_gemm_gpu.py tests/test_sweep_gpu.py tests/test_member_coverage_gpu.py tests/test_workspace_gpu.py -q -m gpu -x > gpurun_out/r04c/tp_tests.txt 2>&1
tail -15 gpurun_out/r04c/tp_tests.txt
