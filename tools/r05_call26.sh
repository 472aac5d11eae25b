#!/bin/bash
o=gpurun_out/r05x; mkdir -p $o
for a in "8192 28672" "12288 8192" "22016 8192" "11008 8192"; do timeout 120 ./tools/stream_lab $a >> $o/stream.txt 2>&1; done
cat $o/stream.txt
