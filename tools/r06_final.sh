#!/bin/bash
# tools/r06_final.sh [tag]: the evidence run of round 6 (one gpurun call): bench.py with default flags (the driver's command) ->
# <tag>/bench.json (last line) + members file; smoke(); rocprofv3 --kernel-trace --stats of the headline step alone, of the Llama-70B
# members the metric names (tools/r06_70b.py) and of the MFMA members; FETCH_SIZE / WRITE_SIZE passes of the step (eager, counters
# only, separate runs); SQ / TCC counters of the M = 4096 members incl. the 28672 x 8192 70B shape (tools/pmc_gemm.sh).
# Copy what should be judged into profiles/.
TAG=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out/$TAG
mkdir -p $out
timeout 900 python bench.py > $out/bench_stdout.txt 2> $out/bench.err
tail -1 $out/bench_stdout.txt > $out/bench.json
wc -c $out/bench.json; cat $out/bench.json
cp gpurun_out/bench_members.json $out/bench_members.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/steptrace -o trace -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-members --no-live-pmc > $out/steptrace_stdout.log 2>&1
f=$(ls $out/steptrace/*/*kernel_stats.csv $out/steptrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f > $out/step_kernel_stats.csv && cat $out/step_kernel_stats.csv
tail -1 $out/steptrace_stdout.log > $out/step_bench.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t70 -o trace -- python $root/tools/r06_70b.py gemv gemm strict > $out/llama70b_stdout.log 2>&1
f=$(ls $out/t70/*/*kernel_stats.csv $out/t70/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f | cut -c1-260 > $out/llama70b_kernel_stats.csv && cat $out/llama70b_kernel_stats.csv
grep -a "^{" $out/llama70b_stdout.log > $out/llama70b_members_under_trace.jsonl
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/gemmtrace -o trace -- python $root/tools/run_gemm_members.py > $out/gemmtrace_stdout.log 2>&1
f=$(ls $out/gemmtrace/*/*kernel_stats.csv $out/gemmtrace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "Name|wqaa::" $f | cut -c1-260 > $out/gemm_kernel_stats.csv && cat $out/gemm_kernel_stats.csv
grep -a "^{" $out/gemmtrace_stdout.log > $out/gemm_members_under_trace.jsonl
rm -rf $out/steptrace $out/gemmtrace $out/t70
cd $root
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc_$ctr -o pmc -- python $root/bench.py --steps 3 --warmup 1 --layers 4 --no-cpu-baseline --no-members --no-live-pmc --eager > $out/pmc_${ctr}_stdout.log 2>&1)
done
python tools/summarize_pmc_dir.py $out 16 > $out/pmc_gemv.json 2>/dev/null; tail -12 $out/pmc_gemv.json
./tools/pmc_gemm.sh f16_u4_m4096 4096 4096 4096 0 4 128 1 0 > $out/pmc_gemm_a.txt 2>&1
./tools/pmc_gemm.sh i8_i2_m4096 4096 4096 4096 1 2 -1 0 3 > $out/pmc_gemm_b.txt 2>&1
./tools/pmc_gemm.sh f16_u4_m4096_n28672k8192 4096 28672 8192 0 4 128 1 0 > $out/pmc_gemm_c.txt 2>&1
./tools/pmc_gemm.sh f16_u4_m128 128 4096 4096 0 4 128 1 0 > $out/pmc_gemm_d.txt 2>&1
python tools/summarize_pmc_gemm.py $root/gpurun_out f16_u4_m4096 i8_i2_m4096 f16_u4_m4096_n28672k8192 f16_u4_m128 > $out/pmc_gemm.json 2>/dev/null
grep -E "mfma_pipe_busy|lds_bank|clock_GHz|launch_ns" $out/pmc_gemm.json
rm -rf $root/gpurun_out/pmc_f16_u4_m4096 $root/gpurun_out/pmc_i8_i2_m4096 $root/gpurun_out/pmc_f16_u4_m4096_n28672k8192 $root/gpurun_out/pmc_f16_u4_m128 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls $out
