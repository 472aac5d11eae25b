mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_decode_persistent_gpu.py tests/test_decode_gpu.py -q -m gpu > gpurun_out/r04c/last_tests.txt 2>&1
tail -3 gpurun_out/r04c/last_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
