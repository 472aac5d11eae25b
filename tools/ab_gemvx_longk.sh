#!/bin/bash
# long-K GEMV shapes: exact-product member forced (WQAA_GEMVX=2 ignores the long-K fence) vs the rounding member
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 4096 11008" "1 4096 4096" "1 11008 4096" "1 8192 28672" "1 4096 14336" "2 4096 11008" "1 2048 11008" "1 1024 28672"; do
  for st in 1 0; do
    r=$(WQAA_GEMVX=2 WQ_STRICT=$st timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(gemvx?_[a-z0-9_]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 grid \2 thr \3: \4 us \5 GB\/s/')
    echo "$shape strict=$st : $r"
  done
done
