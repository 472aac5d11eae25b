#!/bin/bash
# skinny / decode members with non-temporal weight loads: M sweep at the bench shapes (compare with the previous build's table)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "16 4096 4096" "32 4096 4096" "64 4096 4096" "16 11008 4096" "32 11008 4096" "64 4096 11008" "128 4096 4096"; do
  for v in 1 0; do
    r=$(WQAA_GEMM_DECODE_LDS=$v timeout 120 ./tools/wq_bench $shape 0 4 128 1 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(tcx[a-z0-9]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.* ([0-9.]+) TFLOP.*/\1 grid \2 thr \3: \4 us \5 TF/')
    echo "$shape lds=$v : $r"
  done
done
