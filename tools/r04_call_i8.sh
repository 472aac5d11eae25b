mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests/test_gemm_pp_gpu.py -q -m gpu -k "dense_int8 or dense_16bit" > gpurun_out/r04c/i8_tests.txt 2>&1
tail -3 gpurun_out/r04c/i8_tests.txt
timeout 900 python tools/r04_ab_two_pass_own.py --int8-only 2>&1 | tee gpurun_out/r04c/ab_two_pass_own_i8.txt
timeout 900 python -m pytest tests/test_two_pass_gpu.py tests/test_gemm_gpu.py tests/test_sweep_gpu.py tests/test_member_coverage_gpu.py tests/test_bitnet_gpu.py -q -m gpu -x > gpurun_out/r04c/tp_tests2.txt 2>&1
tail -4 gpurun_out/r04c/tp_tests2.txt
