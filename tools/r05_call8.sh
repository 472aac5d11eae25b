#!/bin/bash
# K-sliced decode form: parity tests, then the A/B against the members it stands in for
o=gpurun_out/r05g; mkdir -p $o
timeout 900 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu > $o/tests.txt 2>&1; tail -15 $o/tests.txt
timeout 600 python tools/r05_ab_kslice.py > $o/ab_kslice.txt 2> $o/ab.err; cat $o/ab_kslice.txt; tail -3 $o/ab.err
