#!/bin/bash
# final tree: the GPU parity suite with achieved margins + plan log (member coverage), then bench.py as the driver runs it
bash tools/parity_margins.sh r05f
o=gpurun_out/r05f
timeout 900 python bench.py > $o/bench_stdout.txt 2> $o/bench.err
tail -1 $o/bench_stdout.txt > $o/bench.json
wc -c $o/bench.json
cp gpurun_out/bench_members.json $o/bench_members.json
python - <<'PY'
import json
m = json.load(open("gpurun_out/r05f/bench_members.json"))["members_summary"]
for k in ("gemm_uint4_m32", "gemm_uint4_m64", "gemm_uint4_m128", "gemm_uint4_m256", "gemm_uint4_m64_n4096k8192", "gemm_uint4_m4096", "gemm_f16_dense_m4096", "gemm_fp8_m4096_o_n8192_k8192"):
    print(k, m.get(k))
PY
