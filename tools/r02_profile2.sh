#!/bin/bash
# tools/r02_profile2.sh: round-2 closing PMC run (one gpurun call): the 256x256 MFMA members (uint4 x fp16 + zeros,
# int2 x int8, e4m3 x e4m3), the M = 128 / M = 16 members after this round's work, the exact-product GEMV on a
# 117 MB matrix and on the headline's two single-launch shapes -> gpurun_out/r02_pmc_gemm.json (copy to profiles/).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
mkdir -p $out
./tools/pmc_gemm.sh f16_u4_m4096 4096 4096 4096 0 4 128 1 0 > $out/pmc2_a.txt 2>&1
./tools/pmc_gemm.sh i8_i2_m4096 4096 4096 4096 1 2 -1 0 3 > $out/pmc2_b.txt 2>&1
./tools/pmc_gemm.sh fp8_m4096_n8192_k8192 4096 8192 8192 6 8 -1 0 5 > $out/pmc2_c.txt 2>&1
./tools/pmc_gemm.sh m128 128 4096 4096 0 4 128 1 0 > $out/pmc2_d.txt 2>&1
./tools/pmc_gemm.sh m16 16 4096 4096 0 4 128 1 0 > $out/pmc2_e.txt 2>&1
WQ_STRICT=0 ./tools/pmc_gemm.sh gvx_big 1 28672 8192 1 4 128 0 0 > $out/pmc2_f.txt 2>&1
WQ_STRICT=0 ./tools/pmc_gemm.sh gvx_4096 1 4096 4096 1 4 128 0 0 > $out/pmc2_g.txt 2>&1
WQ_STRICT=0 ./tools/pmc_gemm.sh gvx_k11008 1 4096 11008 1 4 128 0 0 > $out/pmc2_h.txt 2>&1
python3 tools/summarize_pmc_gemm.py $out f16_u4_m4096 i8_i2_m4096 fp8_m4096_n8192_k8192 m128 m16 gvx_big gvx_4096 gvx_k11008 > $out/r02_pmc_gemm.json
python3 - $out/r02_pmc_gemm.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["members"].items():
    print(k, v.get("kernel", "")[:90], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in (v.get("derived") or {}).items()})
PY
