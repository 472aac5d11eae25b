mkdir -p gpurun_out/r04c
timeout 600 python tools/r04_decode_longk_probe.py 2>&1 | tee gpurun_out/r04c/decode_longk_probe.txt
