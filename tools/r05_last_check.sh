#!/bin/bash
# the last check of the round on a fresh box: the GPU parity suite with margins and member coverage, bench.py as the driver runs it, smoke()
bash tools/parity_margins.sh r05last
o=gpurun_out/r05last
timeout 900 python bench.py > $o/bench_stdout.txt 2> $o/bench.err
tail -1 $o/bench_stdout.txt > $o/bench.json; wc -c $o/bench.json
cp gpurun_out/bench_members.json $o/bench_members.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
