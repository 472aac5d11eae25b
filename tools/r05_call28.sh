#!/bin/bash
o=gpurun_out/r05aa; mkdir -p $o
timeout 600 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu > $o/tests.txt 2>&1; tail -4 $o/tests.txt
