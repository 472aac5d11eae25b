#!/bin/bash
# tools/parity_margins.sh: the whole GPU parity suite with (a) the achieved error of every floating-point case written out
# (tests/helpers.record_margin -> profiles/rNN_parity_margins.txt) and (b) the plan log tools/member_coverage.py reads
# (which member classes the run exercised).  One gpurun call; the outputs land in gpurun_out/<tag>/.
tag=${1:-r04f}
o=gpurun_out/$tag
mkdir -p $o
rm -f $o/parity_margins.txt $o/plan_log.txt
export WQAA_PARITY_MARGINS=$PWD/$o/parity_margins.txt WQAA_PLAN_LOG=$PWD/$o/plan_log.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $o/gpu_tests.txt 2>&1
tail -5 $o/gpu_tests.txt
sort -o $o/parity_margins.txt $o/parity_margins.txt
python tools/member_coverage.py $o/plan_log.txt > $o/member_coverage.txt 2>&1
tail -4 $o/member_coverage.txt
