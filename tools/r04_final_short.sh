#!/bin/bash
# tools/r04_final_short.sh [tag]: bench.py with default flags, smoke(), and the rocprofv3 kernel traces of the headline step and the
# large-M members (the counter passes of tools/r04_final.sh are not repeated: the members they describe did not change)
TAG=${1:-r04s}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out/$TAG
mkdir -p $out
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/steptrace -o trace -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-members --no-live-pmc > $out/steptrace_stdout.log 2>&1
f=$(ls $out/steptrace/*/*kernel_stats.csv $out/steptrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/step_kernel_stats.csv && cat $out/step_kernel_stats.csv
grep -a "^{\"metric\"\|^{\"members\"" $out/steptrace_stdout.log | tail -1 > $out/step_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/gemmtrace -o trace -- python $root/tools/run_gemm_members.py > $out/gemmtrace_stdout.log 2>&1
f=$(ls $out/gemmtrace/*/*kernel_stats.csv $out/gemmtrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/gemm_kernel_stats.csv && cat $out/gemm_kernel_stats.csv | cut -c1-200
rm -rf $out/steptrace $out/gemmtrace
