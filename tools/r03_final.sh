#!/bin/bash
# tools/r03_final.sh: last evidence run of the round (one gpurun call): the whole GPU parity suite with the plan log on (member
# coverage), then bench.py with default flags, then the kernel stats of the headline step and of the chained step.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
mkdir -p $out
rm -f $out/r03e_plan_log.txt
WQAA_PLAN_LOG=$out/r03e_plan_log.txt timeout 1500 python -m pytest tests -q -m gpu -x > $out/r03e_pytest.log 2>&1
tail -3 $out/r03e_pytest.log
timeout 600 python bench.py > $out/r03e_bench.json 2> $out/r03e_bench.err
tail -c 1500 $out/r03e_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03e_steptrace -o trace -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-members > $out/r03e_steptrace_stdout.log 2>&1
f=$(ls $out/r03e_steptrace/*/*kernel_stats.csv $out/r03e_steptrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/r03e_step_kernel_stats.csv && cat $out/r03e_step_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03e_chaintrace -o trace -- python $root/tools/run_chain.py > $out/r03e_chaintrace_stdout.log 2>&1
f=$(ls $out/r03e_chaintrace/*/*kernel_stats.csv $out/r03e_chaintrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -12 $f > $out/r03e_chain_kernel_stats.csv && cat $out/r03e_chain_kernel_stats.csv
rm -rf $out/r03e_steptrace $out/r03e_chaintrace
