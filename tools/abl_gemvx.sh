#!/bin/bash
# ablation ladder of the exact-product int4 GEMV (lab members, wrong results by construction): where does the time go?
#   ABL bits: 1 = no decode/dot (one XOR per load), 2 = no activation staging / barrier, 4 = no wave reduction / store
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 4096 4096" "1 11008 4096" "1 4096 11008" "1 28672 8192"; do
  for rep in 1 2; do
    for abl in 0 8 16 32 48; do
      r=$(WQAA_GEMVX_R=2 WQAA_GEMVX_ABL=$abl WQ_STRICT=0 timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -1 | sed -E 's/.*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 us \2 GB\/s/')
      echo "$shape abl=$abl : $r"
    done
  done
done
