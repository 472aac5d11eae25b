mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_decode_persistent_gpu.py -q -m gpu > gpurun_out/r04c/decode11_tests.txt 2>&1
tail -5 gpurun_out/r04c/decode11_tests.txt
timeout 600 python tools/r04_ab_decode_persistent.py 2>&1 | tee gpurun_out/r04c/ab_decode_persistent11.txt
timeout 300 python tools/r04_decode_batch_probe2.py 2>&1 | tee gpurun_out/r04c/decode_batch_probe2.txt
