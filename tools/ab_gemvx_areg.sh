#!/bin/bash
# exact-product GEMV: activations staged in LDS (areg=0) vs kept in the lane's registers (areg=1), K = 4096 shapes
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 4096 4096" "1 11008 4096" "1 12288 4096" "1 2048 4096" "1 1024 4096" "1 28672 4096"; do
  for rep in 1 2; do
    for ar in 0 1; do
      r=$(WQAA_GEMVX_AREG=$ar WQ_STRICT=0 timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(gemvx?_[a-z0-9_]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 grid \2 thr \3: \4 us \5 GB\/s/')
      echo "$shape areg=$ar : $r"
    done
  done
done
