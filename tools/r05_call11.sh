#!/bin/bash
o=gpurun_out/r05j; mkdir -p $o
for a in "8192 28672" "4096 11008" "4096 4096" "12288 8192"; do timeout 120 ./tools/granule_lab $a >> $o/granule.txt 2>&1; done
cat $o/granule.txt
