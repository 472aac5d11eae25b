#!/bin/bash
# tools/r03_first_call.sh: what was queued at the end of round 2 without GPU minutes, in one gpurun call:
#   gpurun --timeout 1500 -- 'bash tools/r03_first_call.sh'
# 1. the queued parity tests (LDS-staged 1-bit / 2-bit GEMV members; resident B_decode)       -> gpurun_out/r03_queued_tests.log
# 2. A/B of the selector rule WQAA_GEMV_DIRECT_FIT=1 (spilling register-resident members)     -> gpurun_out/r03_ab_direct_fit.txt
# 3. A/B of the grid cap of the SGPR-bound two-row exact-product members (3 vs 4 resident workgroups per CU)
#                                                                                              -> gpurun_out/r03_ab_gemvx_grid.txt
# 4. bench.py (carries members.gemm_uint4_m4096_resident_decode, gemv_f16_yardstick_*)        -> gpurun_out/r03_bench.json
# 5. (R03_COVERAGE=1) the whole parity suite with WQAA_PLAN_LOG: which member classes it exercises
#                                                                                              -> gpurun_out/r03_member_coverage.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
WQAA_TEST_NEXT=1 timeout 900 python -m pytest tests/test_gemv_gpu.py tests/test_zz_decoded_cache_gpu.py -q -m gpu -k "low_bit or decoded_cache" > $out/r03_queued_tests.log 2>&1
tail -5 $out/r03_queued_tests.log
timeout 600 python tools/ab_direct_fit.py > $out/r03_ab_direct_fit.txt 2>&1
cat $out/r03_ab_direct_fit.txt
# q/k/v-sized and gate/up-sized single operators of the headline step: 768 / 1024 / uncapped workgroups
timeout 600 python tools/ab_knobs.py "12288 4096" "22016 4096" "11008 4096" -- "WQAA_GEMVX_GRID=768" "WQAA_GEMVX_GRID=1024" "WQAA_GEMVX_GRID=1536" "WQAA_GEMV_UNCAP=1" > $out/r03_ab_gemvx_grid.txt 2>&1
cat $out/r03_ab_gemvx_grid.txt
timeout 900 python bench.py > $out/r03_bench.json 2> $out/r03_bench.err
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03_bench.json")))
print("headline", d["value"], d["roofline"]["frac"])
for k in ("gemm_uint4_m4096", "gemm_uint4_m4096_tuned", "gemm_uint4_m4096_resident_decode"):
    print(k, {x: d["members"][k].get(x) for x in ("us_per_launch", "TFLOPs", "kernel", "error")})
PY
if [ -n "$R03_COVERAGE" ]; then
  rm -f $out/r03_plan.log
  WQAA_PLAN_LOG=$out/r03_plan.log timeout 1500 python -m pytest tests -m gpu -q -x > $out/r03_suite.log 2>&1
  tail -3 $out/r03_suite.log
  python tools/member_coverage.py $out/r03_plan.log > $out/r03_member_coverage.txt 2>&1
  head -60 $out/r03_member_coverage.txt
fi
