#!/bin/bash
# M = 128, 4096^2 uint4 + zeros: main kernel vs reduce kernel time (rocprofv3 kernel trace) over the split-K count
cd ${GRAFT_REPO_ROOT:-.}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ks in 1 2 4 8 16; do
  rm -rf /tmp/m128_$ks
  WQAA_GEMM_KSPLIT=$ks rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/m128_$ks -o t -- $root/tools/wq_bench 128 4096 4096 0 4 128 1 0 3 0 > /tmp/m128_$ks.log 2>&1
  f=$(ls /tmp/m128_$ks/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== ksplit=$ks: $(grep graph: /tmp/m128_$ks.log | sed -E 's/.*graph: ([0-9.]+) us.*/\1 us per launch (graph)/')"
  [ -n "$f" ] && python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:3]:
    print('   ', r['Name'].split('<')[0].replace('void wqaa::',''), 'calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
done
