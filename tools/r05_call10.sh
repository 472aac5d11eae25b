#!/bin/bash
o=gpurun_out/r05i; mkdir -p $o
timeout 300 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu -k "11008 or 28672" > $o/tests0.txt 2>&1; tail -2 $o/tests0.txt
WQAA_GEMM_KSL_MAP=1 timeout 300 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu -k "11008 or 28672" > $o/tests1.txt 2>&1; tail -2 $o/tests1.txt
timeout 600 python tools/r05_ab_kslice.py > $o/ab_kslice.txt 2> $o/ab.err; cat $o/ab_kslice.txt; tail -3 $o/ab.err
