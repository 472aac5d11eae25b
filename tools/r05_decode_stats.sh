#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r05ds; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python $root/tools/run_decode_members.py > $out/stdout.log 2>&1
f=$(ls $out/trace/*/*kernel_stats.csv $out/trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f | cut -c1-260 > $out/decode_kernel_stats.csv && cat $out/decode_kernel_stats.csv
grep -a "^{" $out/stdout.log
rm -rf $out/trace
