mkdir -p gpurun_out/r04c
o=gpurun_out/r04c/gemv_norm_timeline.txt
: > $o
for args in "4096 4096 --count 3" "4096 4096 --count 3 --pro 3" "11008 4096 --pro 2" "11008 4096 --pro 4" "4096 4096" ; do
  echo "== tools/gemv_lab $args" >> $o
  timeout 120 tools/gemv_lab $args >> $o 2>&1
  echo "rc=$?" >> $o
done
tail -5 $o
