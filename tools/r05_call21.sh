#!/bin/bash
# final tree: the GPU parity suite with achieved margins + plan log (member coverage), then the evidence run (bench, smoke, rocprofv3 stats, PMC)
bash tools/parity_margins.sh r05z
bash tools/r05_final.sh r05z > gpurun_out/r05z/final_stdout.txt 2>&1
tail -40 gpurun_out/r05z/final_stdout.txt
