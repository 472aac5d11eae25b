#!/bin/bash
o=gpurun_out/r05p; mkdir -p $o
timeout 600 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu > $o/tests.txt 2>&1; tail -3 $o/tests.txt
timeout 600 python tools/r05_ab_kslice.py quick > $o/ab_kslice.txt 2> $o/ab.err; cat $o/ab_kslice.txt; tail -3 $o/ab.err
timeout 120 ./tools/decode_trace 8 8192 28672 -1 2>&1 | tail -14
timeout 120 ./tools/decode_trace 8 8192 16384 -1 2>&1 | tail -4
