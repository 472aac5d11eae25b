#!/bin/bash
o=gpurun_out/r05last; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu -x > $o/gpu_tests.txt 2>&1; tail -3 $o/gpu_tests.txt
timeout 600 python bench.py > $o/bench_stdout.txt 2> $o/bench.err; tail -1 $o/bench_stdout.txt | cut -c1-400; tail -1 $o/bench_stdout.txt | wc -c
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
