#!/bin/bash
# round 5, GPU call 1: the mid-M member's parity tests + same-process A/B, the DMA lab's M0 variants, the bench line as the driver sees it
o=gpurun_out/r05c1
mkdir -p $o
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_mid_gpu.py -x -q > $o/test_mid.txt 2>&1; echo "test_mid rc=$?" | tee -a $o/summary.txt
tail -5 $o/test_mid.txt | tee -a $o/summary.txt
timeout 600 python tools/r05_ab_mid.py > $o/ab_mid.txt 2>&1; echo "ab_mid rc=$?" | tee -a $o/summary.txt
cat $o/ab_mid.txt | tee -a $o/summary.txt
timeout 300 tools/dma_lab > $o/lab_dma_stream.txt 2>&1; echo "dma_lab rc=$?" | tee -a $o/summary.txt
tail -70 $o/lab_dma_stream.txt | tee -a $o/summary.txt
timeout 600 python bench.py > $o/bench_stdout.txt 2> $o/bench_stderr.txt; echo "bench rc=$?" | tee -a $o/summary.txt
tail -1 $o/bench_stdout.txt | wc -c | tee -a $o/summary.txt
tail -1 $o/bench_stdout.txt | tee -a $o/summary.txt
cp gpurun_out/bench_members.json $o/ 2>/dev/null
