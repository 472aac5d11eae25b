mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_decode_persistent_gpu.py -q -m gpu > gpurun_out/r04c/long_tests2.txt 2>&1
tail -15 gpurun_out/r04c/long_tests2.txt
timeout 600 python tools/r04_ab_decode_long.py 2>&1 | tee gpurun_out/r04c/ab_decode_long2.txt
