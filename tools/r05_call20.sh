#!/bin/bash
o=gpurun_out/r05s; mkdir -p $o
timeout 900 python -m pytest tests/test_gemm_kslice_gpu.py tests/test_gemm_gpu.py tests/test_gemm_tail_gpu.py -x -q -m gpu > $o/tests.txt 2>&1; tail -3 $o/tests.txt
timeout 600 python tools/r05_ab_zint.py > $o/ab_zint.txt 2> $o/ab.err; cat $o/ab_zint.txt; tail -3 $o/ab.err
