#!/bin/bash
o=gpurun_out/r05c5
mkdir -p $o
timeout 1500 python -m pytest tests/test_gemm_pp_gpu.py tests/test_c5_gpu.py tests/test_two_pass_gpu.py tests/test_chain_gpu.py tests/test_workspace_gpu.py -x -q > $o/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $o/tests.txt
timeout 900 python tools/r05_ab_dense_wide.py > $o/ab_dense_wide.txt 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $o/ab_dense_wide.txt
