#!/bin/bash
o=gpurun_out/r05y; mkdir -p $o
for a in "4096 4096 g" "12288 4096 g" "22016 4096 g" "4096 11008" "12288 8192 g" "8192 28672 g"; do timeout 120 ./tools/stream_lab $a 2>&1 | head -5 >> $o/stream_dyn.txt; done
cat $o/stream_dyn.txt
