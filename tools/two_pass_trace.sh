#!/bin/bash
# rocprofv3 kernel trace of the two-pass member at uint4 4096^3 (dequant kernel + the library's GEMM kernel)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tp_trace -o trace -- python $root/tools/ab_two_pass.py > $out/tp_trace_stdout.log 2>&1
f=$(ls $out/tp_trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|dequant|Cijk|wq_gemm_kernel" $f | cut -c1-200 | head -20
rm -rf $out/tp_trace/*kernel_trace.csv
