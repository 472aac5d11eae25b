#!/bin/bash
# same-call A/B: skinny (4 k-steps in flight, split-K by 4 steps) vs pipelined member for M = 65..256
cd $GRAFT_REPO_ROOT
for shape in "96 4096 4096" "128 4096 4096" "192 4096 4096" "256 4096 4096" "128 11008 4096" "128 4096 11008"; do
  for v in "64 0" "512 0" "512 4" "512 2" "64 0" "512 0" "512 4" "512 2"; do
    set -- $v
    if [ "$2" = "0" ]; then
      r=$(WQAA_GEMM_SKINNY_MAXM=$1 timeout 60 ./tools/wq_bench $shape 0 4 128 1 0 3 | sed -n '1p;$p' | tr '\n' ' ' | sed -E 's/^(\S+) .*graph: ([0-9.]+) us.*/\1 \2 us/')
    else
      r=$(WQAA_GEMM_SKINNY_MAXM=$1 WQAA_GEMM_MF=$2 timeout 60 ./tools/wq_bench $shape 0 4 128 1 0 3 | sed -n '1p;$p' | tr '\n' ' ' | sed -E 's/^(\S+) .*graph: ([0-9.]+) us.*/\1 \2 us/')
    fi
    echo "$shape skinny_max=$1 mf=$2 : $r"
  done
done
