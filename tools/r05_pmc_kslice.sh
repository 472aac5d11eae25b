#!/bin/bash
# counters of the K-sliced decode form (and, for contrast, the whole-tile form at 12288 x 8192): tools/pmc_gemm.sh, separate passes
o=gpurun_out/r05pmc; mkdir -p $o
./tools/pmc_gemm.sh f16_u4_m8_n8192k28672 8 8192 28672 0 4 128 1 0 > $o/pmc_kslice.txt 2>&1
./tools/pmc_gemm.sh f16_u4_m8_n12288k8192 8 12288 8192 0 4 128 1 0 > $o/pmc_xdlt.txt 2>&1
python tools/summarize_pmc_gemm.py $(pwd)/gpurun_out f16_u4_m8_n8192k28672 f16_u4_m8_n12288k8192 > $o/pmc_decode.json 2>/dev/null
grep -E "kernel|launch_ns|clock_GHz|mfma_pipe_busy|TCC_EA_RDREQ|hbm|wait" $o/pmc_decode.json | head -40
rm -rf gpurun_out/pmc_f16_u4_m8_n8192k28672 gpurun_out/pmc_f16_u4_m8_n12288k8192
