#!/bin/bash
# A/B in ONE call: M <= 16 decode-batch member, activations as fragment-shaped global loads (201) vs through LDS-DMA (211)
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_te_golden.py tests/test_sweep_gpu.py -q -x 2>&1 | tail -5
for shape in "16 4096 4096" "12 4096 4096" "8 4096 4096" "3 4096 4096" "16 3584 8192"; do
  for rep in 1 2; do
    for v in 0 1; do
      r=$(WQAA_GEMM_DECODE_LDS=$v timeout 120 ./tools/wq_bench $shape 0 4 128 1 0 5 1 | tail -2 | tr '\n' ' ' | sed -E 's/.*(tcx[a-z0-9]+).*grid=([0-9]+) threads=([0-9]+).*graph: ([0-9.]+) us.* ([0-9.]+) TFLOP.*/\1 grid \2 thr \3: \4 us \5 TF/')
      echo "$shape lds=$v : $r"
    done
  done
done
# int2 x int8
for v in 0 1; do
  WQAA_GEMM_DECODE_LDS=$v timeout 120 ./tools/wq_bench 16 4096 4096 1 2 -1 0 3 5 1 2>&1 | tail -2 | tr '\n' ' '; echo " lds=$v"
done
