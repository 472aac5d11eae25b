#!/bin/bash
# tools/r06_suite.sh [tag]: the whole GPU parity suite ONCE under rocprofv3 --kernel-trace --stats (the kernel census of tools/kernel_census.py),
# with the achieved margins (tests/helpers.record_margin) and the plan log (tools/member_coverage.py) written out.  One gpurun call.
tag=${1:-r06s}
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out/$tag
mkdir -p $o
rm -f $o/parity_margins.txt $o/plan_log.txt
export WQAA_PARITY_MARGINS=$o/parity_margins.txt WQAA_PLAN_LOG=$o/plan_log.txt
cd /tmp && export TMPDIR=/tmp
timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o suite_%pid% -- python -m pytest $root/tests -q -m gpu -p no:cacheprovider --rootdir $root --durations=25 > $o/gpu_tests.txt 2>&1
cd $root
grep -E "passed|failed|error" $o/gpu_tests.txt | tail -3
# (one CSV per traced process: the two-rank tests spawn their own)
i=0; for f in $(ls $o/trace/*/*kernel_stats.csv $o/trace/*kernel_stats.csv 2>/dev/null); do cp $f $o/suite_kernel_stats_$i.csv; i=$((i+1)); done
rm -rf $o/trace
sort -o $o/parity_margins.txt $o/parity_margins.txt
python tools/member_coverage.py $o/plan_log.txt > $o/member_coverage.txt 2>&1
head -4 $o/member_coverage.txt | cut -c1-200
python tools/kernel_census.py --list $o/suite_kernel_stats_*.csv > $o/kernel_census.txt 2>&1
grep -v "debug_decode" $o/kernel_census.txt | head -30
