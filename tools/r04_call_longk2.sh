mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_decode_persistent_gpu.py tests/test_gemm_gpu.py tests/test_member_coverage_gpu.py -q -m gpu -x > gpurun_out/r04c/longk2_tests.txt 2>&1
tail -4 gpurun_out/r04c/longk2_tests.txt
timeout 600 python tools/r04_decode_longk_probe.py 2>&1 | tee gpurun_out/r04c/decode_longk_probe2.txt
timeout 300 python tools/r04_decode_batch_probe2.py 2>&1 | tee gpurun_out/r04c/decode_batch_probe3.txt
