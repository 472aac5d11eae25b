#!/bin/bash
# A/B in ONE call: strict (per-element rounding) members vs the exact-product members (WQ_STRICT=0), int4 g128 + scale
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 4096 4096" "1 11008 4096" "1 4096 11008" "1 12288 4096" "1 28672 8192" "1 8192 28672" "1 1024 28672" "1 1280 8192" "1 1024 1024" "2 4096 4096" "2 11008 4096"; do
  for rep in 1 2; do
    for st in 1 0; do
      r=$(WQ_STRICT=$st timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(gemvx?_[a-z0-9_]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 grid \2 thr \3: \4 us \5 GB\/s/')
      echo "$shape strict=$st : $r"
    done
  done
done
echo "--- tile knobs (exact members) ---"
for shape in "1 4096 4096" "1 11008 4096" "1 4096 11008"; do
  for knobs in "WQAA_GEMVX_R=1" "WQAA_GEMVX_R=2" "WQAA_GEMVX_R=2 WQAA_GEMVX_KW=2" "WQAA_GEMVX_SLOTS=4" "WQAA_GEMVX_SLOTS=16" "WQAA_GEMVX_R=1 WQAA_GEMVX_SLOTS=16"; do
    r=$(env $knobs WQ_STRICT=0 timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(gemvx_[a-z0-9]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 grid \2 thr \3: \4 us \5 GB\/s/')
    echo "$shape $knobs : $r"
  done
done
