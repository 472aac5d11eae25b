mkdir -p gpurun_out/r04c
o=gpurun_out/r04c/ab_kernarg.txt
: > $o
for v in unset 0 1 unset 1 0; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> $o
  timeout 300 python bench.py --no-members --no-cpu-baseline --no-live-pmc --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({k:r[k] for k in ('value','ms_per_step')}), r['roofline']['frac'])" >> $o
  timeout 100 tools/gemv_lab 4096 4096 2>&1 | grep -E "library launch|first weight loads issued|wave start" | head -3 >> $o
done
cat $o
