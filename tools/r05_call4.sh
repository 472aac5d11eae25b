#!/bin/bash
o=gpurun_out/r05c4
mkdir -p $o
timeout 1200 python -m pytest tests/test_gemm_mid_gpu.py tests/test_gemm_gpu.py tests/test_gemm_pp_gpu.py tests/test_gemm_tail_gpu.py -x -q > $o/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $o/tests.txt
timeout 900 python tools/r05_ab_mid.py > $o/ab_mid.txt 2>&1; echo "ab_mid rc=$?"
grep -v amdgpu.ids $o/ab_mid.txt
