#!/bin/bash
o=gpurun_out/r05l; mkdir -p $o
for a in "8192 28672" "8192 16384" "12288 8192" "4096 8192" "4096 4096"; do timeout 120 ./tools/stream_lab $a >> $o/stream.txt 2>&1; done
cat $o/stream.txt
