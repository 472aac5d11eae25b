#!/bin/bash
o=gpurun_out/r05c3
mkdir -p $o
timeout 900 python -m pytest tests/test_gemm_mid_gpu.py -x -q > $o/test_mid.txt 2>&1; echo "test_mid rc=$?"; tail -4 $o/test_mid.txt
for a in "128 4096 4096 -1" "128 4096 4096" "64 4096 4096 -1" "32 4096 4096 -1"; do
  echo "== mid_trace $a" >> $o/mid_trace.txt
  timeout 120 tools/mid_trace $a >> $o/mid_trace.txt 2>&1
done
cat $o/mid_trace.txt
timeout 900 python tools/r05_ab_mid.py > $o/ab_mid.txt 2>&1; echo "ab_mid rc=$?"
grep -v amdgpu.ids $o/ab_mid.txt
