#!/bin/bash
o=gpurun_out/r05c2
mkdir -p $o
for a in "128 4096 4096" "128 4096 4096 -1" "64 4096 4096" "32 4096 4096"; do
  echo "== mid_trace $a" >> $o/mid_trace.txt
  timeout 120 tools/mid_trace $a >> $o/mid_trace.txt 2>&1
done
cat $o/mid_trace.txt
timeout 900 python tools/r05_ab_mid.py > $o/ab_mid.txt 2>&1; echo "ab_mid rc=$?"
grep -v amdgpu.ids $o/ab_mid.txt
