#!/bin/bash
o=gpurun_out/r05u; mkdir -p $o
for a in "8 8192 28672 -1" "8 12288 8192 -1"; do
  echo "=== decode_trace $a" >> $o/trace.txt
  timeout 120 ./tools/decode_trace $a 2>&1 | tail -16 >> $o/trace.txt
done
cat $o/trace.txt
