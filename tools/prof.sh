#!/bin/bash
# tools/prof.sh <tag> <cmd...>: run <cmd> under rocprofv3 --kernel-trace --stats (CSV) on the GPU box and
# leave a compact per-kernel summary in gpurun_out/prof_<tag>.txt (copy what matters into profiles/).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- "$@" > $out/stdout.log 2>&1
f=$(ls $out/*kernel_stats.csv 2>/dev/null | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- $*"; grep -v "^\[" $out/stdout.log | tail -20; echo; [ -n "$f" ] && head -25 "$f"; } > $root/gpurun_out/prof_$tag.txt
rm -f $out/*kernel_trace.csv $out/*agent_info.csv 2>/dev/null
cat $root/gpurun_out/prof_$tag.txt
