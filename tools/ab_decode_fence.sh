#!/bin/bash
# where does the decode-LDS member (one 16-row weight fragment per workgroup, 1 workgroup per CU) beat the skinny split-K member?
cd ${GRAFT_REPO_ROOT:-.}
for shape in "16 2048 4096" "16 3072 4096" "16 5120 4096" "16 8192 4096" "16 11008 4096" "16 4096 11008" "16 2048 8192" "16 8192 8192" "8 6144 4096" "8 8192 4096" "8 11008 4096" "8 4096 11008" "4 11008 4096" "4 8192 8192"; do
  for v in 1 0; do
    r=$(WQAA_GEMM_DECODE_FORCE=$v timeout 120 ./tools/wq_bench $shape 0 4 128 1 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(tcx[a-z0-9]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.* ([0-9.]+) TFLOP.*/\1 grid \2 thr \3: \4 us \5 TF/')
    echo "$shape force=$v : $r"
  done
done
