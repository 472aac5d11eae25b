#!/bin/bash
o=gpurun_out/r05suite
mkdir -p $o
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $o/suite.txt 2>&1; echo "suite rc=$?"
tail -25 $o/suite.txt
