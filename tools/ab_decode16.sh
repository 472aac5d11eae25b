#!/bin/bash
# same-call A/B: decode-batch member vs skinny split-K member for M = 9..16
cd $GRAFT_REPO_ROOT
for shape in "12 4096 4096" "16 4096 4096" "16 4096 11008" "16 2048 8192" "16 5120 5120" "10 4096 4096"; do
  for cfg in "0 4 128 1 0" "1 2 -1 0 3"; do
    for mm in 8 16 8 16 8 16; do
      r=$(WQAA_GEMM_DECODE_MAXM=$mm timeout 60 ./tools/wq_bench $shape $cfg 3 | tail -1 | sed -E 's/.*graph: ([0-9.]+) us.*/\1 us/')
      echo "$shape [$cfg] maxm=$mm : $r"
    done
  done
done
