mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_gemm_pp_gpu.py -q -m gpu -k "dense_16bit or dense_fp8" > gpurun_out/r04c/dense_tests.txt 2>&1
tail -4 gpurun_out/r04c/dense_tests.txt
timeout 900 python tools/r04_dense16.py 2>&1 | tee gpurun_out/r04c/dense16.txt
