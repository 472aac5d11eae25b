#!/bin/bash
o=gpurun_out/r05q; mkdir -p $o
for a in "8 8192 28672 -1" "16 8192 28672 -1" "8 4096 11008 -1"; do
  echo "=== decode_trace $a" >> $o/trace.txt
  timeout 120 ./tools/decode_trace $a 2>&1 | tail -13 >> $o/trace.txt
done
cat $o/trace.txt
