root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r04s
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dectrace -o trace -- python $root/tools/run_decode_members.py > $out/dectrace_stdout.log 2>&1
f=$(ls $out/dectrace/*/*kernel_stats.csv $out/dectrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/decode_kernel_stats.csv && cut -c1-220 $out/decode_kernel_stats.csv
grep -a "^{" $out/dectrace_stdout.log
rm -rf $out/dectrace
