mkdir -p gpurun_out/r04c
o=gpurun_out/r04c
timeout 900 python -m pytest tests/test_gemm_tail_gpu.py tests/test_bitnet_gpu.py tests/test_parallel_gpu.py tests/test_gemm_pp_gpu.py -q -m gpu -rs > $o/call4_tests.txt 2>&1
tail -12 $o/call4_tests.txt
timeout 900 python tools/r04_ab_gemm.py > $o/ab_gemm2.txt 2>&1
cat $o/ab_gemm2.txt
