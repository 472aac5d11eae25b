#!/bin/bash
# tools/profile_round.sh <tag>: the round's evidence run on the GPU box.
#   1. bench.py (default flags)                          -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same bench -> gpurun_out/<tag>_kernel_stats.csv
#   2b. the same for the headline step alone (--no-members) -> gpurun_out/<tag>_step_kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (own runs, eager launches, no other tracing)
# Copy what should be judged from gpurun_out/ into profiles/.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 3000 $out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o trace -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/${tag}_trace_stdout.log 2>&1
f=$(ls $out/${tag}_trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv && head -12 $out/${tag}_kernel_stats.csv
rm -rf $out/${tag}_trace/*kernel_trace.csv
# 2b. the same trace of the headline step alone (no members): every GEMV launch in it belongs to the step
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_steptrace -o trace -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-members > $out/${tag}_steptrace_stdout.log 2>&1
f=$(ls $out/${tag}_steptrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/${tag}_step_kernel_stats.csv && cat $out/${tag}_step_kernel_stats.csv
grep -a "^{\"metric\"" $out/${tag}_steptrace_stdout.log | tail -1 > $out/${tag}_step_bench.json
rm -rf $out/${tag}_steptrace/*kernel_trace.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $out/${tag}_pmc_$ctr -o pmc -- python $root/bench.py --steps 3 --warmup 1 --layers 4 --no-cpu-baseline --no-members --eager > $out/${tag}_pmc_${ctr}_stdout.log 2>&1
  ls $out/${tag}_pmc_$ctr | head
done
python $root/tools/summarize_pmc.py $out $tag 16
