#!/bin/bash
# tools/r03_final.sh: last evidence run of the round (one gpurun call): the whole GPU parity suite with the plan log on (member
# coverage), then bench.py with default flags, then the kernel stats of the headline step and of the chained step.
TAG=${R03_TAG:-r03e}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
mkdir -p $out
rm -f $out/${TAG}_plan_log.txt
WQAA_PLAN_LOG=$out/${TAG}_plan_log.txt timeout 1500 python -m pytest tests -q -m gpu -x > $out/${TAG}_pytest.log 2>&1
tail -3 $out/${TAG}_pytest.log
timeout 600 python bench.py > $out/${TAG}_bench.json 2> $out/${TAG}_bench.err
tail -c 1500 $out/${TAG}_bench.json
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${TAG}_steptrace -o trace -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-members > $out/${TAG}_steptrace_stdout.log 2>&1
f=$(ls $out/${TAG}_steptrace/*/*kernel_stats.csv $out/${TAG}_steptrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/${TAG}_step_kernel_stats.csv && cat $out/${TAG}_step_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${TAG}_chaintrace -o trace -- python $root/tools/run_chain.py > $out/${TAG}_chaintrace_stdout.log 2>&1
f=$(ls $out/${TAG}_chaintrace/*/*kernel_stats.csv $out/${TAG}_chaintrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -12 $f > $out/${TAG}_chain_kernel_stats.csv && cat $out/${TAG}_chain_kernel_stats.csv
rm -rf $out/${TAG}_steptrace $out/${TAG}_chaintrace
