#!/bin/bash
o=gpurun_out/r05m; mkdir -p $o
timeout 900 python tools/r05_ab_kslice.py > $o/ab_kslice.txt 2> $o/ab.err; cat $o/ab_kslice.txt; tail -3 $o/ab.err
