cd $GRAFT_REPO_ROOT
for M in 16 64 128 256 512 1024 2048; do
  for mf in 1 2 4 8 16; do
    bm=$((mf*16)); if [ $bm -gt $((M*2)) ]; then continue; fi
    if [ $mf -le 2 ] && [ $M -ge 256 ]; then continue; fi
    for ks in 1 2 4 8 16; do
      r=$(WQAA_GEMM_MF=$mf WQAA_GEMM_KSPLIT=$ks timeout 60 ./tools/wq_bench $M 4096 4096 0 4 128 1 0 2 | tail -1 | sed -E 's/.*graph: ([0-9.]+) us.* ([0-9.]+) TFLOP.*/\1 us \2 TF/')
      echo "M=$M mf=$mf ks=$ks : $r"
    done
  done
done
