#!/bin/bash
# tools/pmc_gemm.sh <tag> <wq_bench args...>: SQ counters of one kernel (eager launches, counters only)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out -o a -- $root/tools/wq_bench "$@" 2 0 > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out -o b -- $root/tools/wq_bench "$@" 2 0 > $out/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out -o c -- $root/tools/wq_bench "$@" 2 0 > $out/c.log 2>&1
rocprofv3 --pmc TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum --output-format csv -d $out -o d -- $root/tools/wq_bench "$@" 2 0 > $out/d.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "wq_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
