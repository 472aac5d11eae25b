mkdir -p gpurun_out/r04c
o=gpurun_out/r04c
timeout 600 python -m pytest tests/test_gemm_tail_gpu.py tests/test_bitnet_gpu.py tests/test_parallel_gpu.py -x -q -m gpu -rs > $o/call3_tests.txt 2>&1
tail -8 $o/call3_tests.txt
timeout 900 python tools/r04_ab_gemm.py > $o/ab_gemm.txt 2>&1
cat $o/ab_gemm.txt
