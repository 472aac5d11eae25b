#!/bin/bash
o=gpurun_out/r05c6
mkdir -p $o
timeout 600 python tools/ab_pp_tile.py f8 "256 8192 8192" "512 8192 8192" "1024 8192 8192" "2048 8192 8192" "3072 8192 8192" "1024 28672 8192" "2048 10240 8192" "4096 3584 8192" "4096 1280 8192" "512 28672 8192" 2>&1 | grep -v amdgpu.ids > $o/ab_pp_tile_f8.txt
cat $o/ab_pp_tile_f8.txt
timeout 400 python tools/ab_pp_tile.py f16 "1024 4096 4096" "2048 4096 4096" "3072 4096 4096" "1536 4096 4096" "2048 11008 4096" "1024 8192 8192" 2>&1 | grep -v amdgpu.ids > $o/ab_pp_tile_f16.txt
cat $o/ab_pp_tile_f16.txt
