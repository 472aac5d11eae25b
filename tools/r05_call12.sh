#!/bin/bash
o=gpurun_out/r05k; mkdir -p $o
WQAA_GEMM_KSL_MAP=1 timeout 300 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu > $o/tests_dma.txt 2>&1; tail -3 $o/tests_dma.txt
WQAA_GEMM_KSL_W=1 timeout 300 python -m pytest tests/test_gemm_kslice_gpu.py -x -q -m gpu > $o/tests_reg.txt 2>&1; tail -3 $o/tests_reg.txt
timeout 900 python tools/r05_ab_kslice.py > $o/ab_kslice.txt 2> $o/ab.err; cat $o/ab_kslice.txt; tail -3 $o/ab.err
for a in "8 4096 11008 -2" "8 8192 28672 -2"; do
  echo "=== decode_trace $a" >> $o/trace.txt
  timeout 120 ./tools/decode_trace $a 2>&1 | tail -11 >> $o/trace.txt
done
cat $o/trace.txt
