tag=r02c; root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_steptrace -o trace -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-members > $out/${tag}_steptrace_stdout.log 2>&1
f=$(ls $out/${tag}_steptrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/${tag}_step_kernel_stats.csv && cat $out/${tag}_step_kernel_stats.csv
grep -a "^{\"metric\"" $out/${tag}_steptrace_stdout.log | tail -1 > $out/${tag}_step_bench.json
rm -rf $out/${tag}_steptrace/*kernel_trace.csv
