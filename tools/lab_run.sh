#!/bin/bash
# one GPU call: the ping-pong member's variants against the shipped member (tools/gemm_lab.hip)
out=gpurun_out/${1:-lab}.txt
shift
: > $out
run() { echo "== $*" | tee -a $out; timeout 150 "$@" 2>&1 | tee -a $out; echo "rc=$?" >> $out; }
if [ $# -gt 0 ]; then
  while [ $# -gt 0 ]; do run $1; shift; done
else
  run tools/gemm_lab 4096 4096 4096 --kind u4
  run tools/gemm_lab 4096 4096 4096 --kind i2
fi
