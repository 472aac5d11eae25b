#!/bin/bash
# tools/r03_profile.sh: round-3 evidence run (one gpurun call): bench + rocprofv3 kernel stats + GEMV HBM counters
# (tools/profile_round.sh r03), then the SQ / TCC counters of the three ping-pong 256x256 members and the mid-M / GEMV
# members (tools/pmc_gemm.sh) -> gpurun_out/r03_pmc_gemm.json.  Copy what should be judged into profiles/.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
mkdir -p $out
bash tools/profile_round.sh ${R03_TAG:-r03} > $out/${R03_TAG:-r03}_profile_round.log 2>&1
./tools/pmc_gemm.sh f16_u4_m4096 4096 4096 4096 0 4 128 1 0 > $out/pmc3_a.txt 2>&1
./tools/pmc_gemm.sh i8_i2_m4096 4096 4096 4096 1 2 -1 0 3 > $out/pmc3_b.txt 2>&1
./tools/pmc_gemm.sh fp8_m4096_n8192_k8192 4096 8192 8192 6 8 -1 0 5 > $out/pmc3_c.txt 2>&1
./tools/pmc_gemm.sh m128 128 4096 4096 0 4 128 1 0 > $out/pmc3_d.txt 2>&1
./tools/pmc_gemm.sh f16_u4_m2048 2048 4096 4096 0 4 128 1 0 > $out/pmc3_e.txt 2>&1
WQ_STRICT=0 ./tools/pmc_gemm.sh gvx_big 1 28672 8192 1 4 128 0 0 > $out/pmc3_f.txt 2>&1
python3 tools/summarize_pmc_gemm.py $out f16_u4_m4096 i8_i2_m4096 fp8_m4096_n8192_k8192 m128 f16_u4_m2048 gvx_big > $out/${R03_TAG:-r03}_pmc_gemm.json
python3 - $out/${R03_TAG:-r03}_pmc_gemm.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["members"].items():
    print(k, v.get("kernel", "")[:90], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in (v.get("derived") or {}).items()})
PY
