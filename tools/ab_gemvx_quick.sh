#!/bin/bash
# quick A/B: strict vs exact members on the headline shapes (+ knob variants of the exact members)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 4096 4096" "1 11008 4096" "1 4096 11008" "1 28672 8192" "2 4096 4096"; do
  for st in 1 0; do
    r=$(WQ_STRICT=$st timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(gemvx?_[a-z0-9_]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 grid \2 thr \3: \4 us \5 GB\/s/')
    echo "$shape strict=$st : $r"
  done
  for knobs in "$@"; do
    r=$(env $knobs WQ_STRICT=0 timeout 120 ./tools/wq_bench $shape 1 4 128 0 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(gemvx_[a-z0-9]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.*-> ([0-9.]+) GB.*/\1 grid \2 thr \3: \4 us \5 GB\/s/')
    echo "$shape $knobs : $r"
  done
done
