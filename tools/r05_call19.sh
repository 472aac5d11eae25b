#!/bin/bash
o=gpurun_out/r05r; mkdir -p $o
timeout 120 ./tools/issue_lab > $o/issue.txt 2>&1; cat $o/issue.txt
