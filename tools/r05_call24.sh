#!/bin/bash
o=gpurun_out/r05v; mkdir -p $o
for n in 0 1 2 4 6 7; do
  for a in "8 8192 28672 -1" "8 22016 4096 0"; do
    echo "=== abl$n decode_trace $a" >> $o/abl.txt
    timeout 120 ./tools/decode_trace_abl$n $a 2>&1 | grep -E "rep 2|launch 63|wave lifetime|all units done|not waiting" | tail -5 >> $o/abl.txt
  done
done
cat $o/abl.txt
