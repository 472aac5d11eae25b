mkdir -p gpurun_out/r04c
o=gpurun_out/r04c
timeout 600 python -m pytest tests/test_parallel_gpu.py tests/test_bitnet_gpu.py tests/test_gemm_pp_gpu.py -x -q -m gpu > $o/call2_tests.txt 2>&1
tail -3 $o/call2_tests.txt
timeout 200 tools/gemm_lab 4096 4096 4096 --kind u4 --rounds 5 > $o/gemm_lab_u4.txt 2>&1
grep "^time" $o/gemm_lab_u4.txt
timeout 200 python tools/run_gemm_members.py > $o/gemm_members.txt 2>&1
cat $o/gemm_members.txt | tail -4
