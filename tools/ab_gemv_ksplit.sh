#!/bin/bash
# rounding GEMV members on few-row shards: K split across the waves of a workgroup (selector's choice) vs none (WQAA_GEMV_KW=1)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 1024 28672" "1 1280 8192" "1 1024 8192" "1 512 11008" "1 2048 11008" "1 3584 8192" "2 1024 28672" "1 4096 4096" "1 11008 4096"; do
  for kw in 1 0; do
    if [ $kw = 0 ]; then pre=""; else pre="WQAA_GEMV_KW=1"; fi
    r=$(env $pre WQ_STRICT=1 timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(gemvx?_[a-z0-9_]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 grid \2 thr \3: \4 us \5 GB\/s/')
    echo "$shape $( [ $kw = 0 ] && echo selector || echo unsplit ) : $r"
  done
done
