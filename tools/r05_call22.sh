#!/bin/bash
o=gpurun_out/r05t; mkdir -p $o
timeout 600 python tools/lab_decode_members.py > $o/lab_forced_zint.txt 2> $o/err.txt; cat $o/lab_forced_zint.txt; tail -2 $o/err.txt
