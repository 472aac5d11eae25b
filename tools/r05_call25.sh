#!/bin/bash
o=gpurun_out/r05w; mkdir -p $o
timeout 600 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu > $o/tests.txt 2>&1; tail -2 $o/tests.txt
timeout 600 python tools/r05_ab_kslice.py quick > $o/ab_kslice.txt 2> $o/ab.err; cat $o/ab_kslice.txt; tail -2 $o/ab.err
timeout 120 ./tools/decode_trace 8 8192 28672 -1 2>&1 | tail -12
