#!/bin/bash
# A/B in ONE call: 4-wave members with per-step 2-byte metadata loads (WQAA_GEMM_WIDE=0) vs the 8-byte block loads
cd $GRAFT_REPO_ROOT
for shape in "128 28672 8192" "256 28672 8192" "384 28672 8192" "128 4096 4096" "256 4096 4096" "512 4096 4096" "1024 4096 4096" "128 8192 8192" "256 8192 8192" "256 11008 4096" "512 4096 11008"; do
  for zm in 0 1; do
    for w in 0 1 0 1 0 1; do
      r=$(WQAA_GEMM_WIDE=$w timeout 120 ./tools/wq_bench $shape 0 4 128 $zm 0 3 | tail -1 | sed -E 's/.*graph: ([0-9.]+) us.* ([0-9.]+) TFLOP.*/\1 us \2 TF/')
      echo "$shape zm=$zm wide=$w : $r"
    done
  done
done
