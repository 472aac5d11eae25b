#!/bin/bash
# tools/r02_profile1.sh: round-2 opening evidence run (one gpurun call): PMC profiles of the kernels VERDICT r01
# names as furthest from their roofline - the mid-M GEMM members (M = 128, M = 16) and the int4 GEMV at three sizes -
# plus a kernel trace of the M = 128 line (main kernel + reduce launch).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
mkdir -p $out
./tools/pmc_gemm.sh m128 128 4096 4096 0 4 128 1 0 > $out/pmc_m128.txt 2>&1
./tools/pmc_gemm.sh m16 16 4096 4096 0 4 128 1 0 > $out/pmc_m16.txt 2>&1
./tools/pmc_gemm.sh gv_big 1 28672 8192 1 4 128 0 0 > $out/pmc_gv_big.txt 2>&1
./tools/pmc_gemm.sh gv_4096 1 4096 4096 1 4 128 0 0 > $out/pmc_gv_4096.txt 2>&1
./tools/pmc_gemm.sh gv_11008 1 11008 4096 1 4 128 0 0 > $out/pmc_gv_11008.txt 2>&1
./tools/pmc_gemm.sh gv_k11008 1 4096 11008 1 4 128 0 0 > $out/pmc_gv_k11008.txt 2>&1
python3 tools/summarize_pmc_gemm.py $out m128 m16 gv_big gv_4096 gv_11008 gv_k11008 > $out/r02_pmc_before.json
cat $out/r02_pmc_before.json | head -150
./tools/prof.sh m128 ./tools/wq_bench 128 4096 4096 0 4 128 1 0 4 1 | tail -12
for s in "1 4096 4096" "1 11008 4096" "1 4096 11008" "1 28672 8192" "128 4096 4096" "16 4096 4096"; do
  zm=0; wf=1; [ "${s%% *}" != "1" ] && { zm=1; wf=0; }
  ./tools/wq_bench $s $wf 4 128 $zm 0 5 1 | tail -1
done
