"""gpurun_out/pmc_<tag>/ (tools/pmc_gemm.sh) -> one JSON with the per-launch counter means of the GEMM kernel
and the figures derived from them (matrix-pipe utilisation, instruction mix per MFMA, wave-state split)."""
import csv, glob, json, sys, collections

def summarize(d):
    acc = collections.defaultdict(list)
    name = dur = None
    durs = []
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "wq_gemm_kernel" in row["Kernel_Name"]:
                name = row["Kernel_Name"]
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    out = {"kernel": name, "counters_mean_per_launch": m, "launch_ns_mean_under_counters": sum(durs) / max(1, len(durs))}
    if "SQ_INSTS_MFMA" in m and "GRBM_GUI_ACTIVE" in m:
        simds, xcds = 1024, 8
        cyc = m["GRBM_GUI_ACTIVE"] / xcds                      # busy cycles of one XCD
        out["derived"] = {
            "cycles_per_launch": cyc,
            "clock_GHz": cyc / out["launch_ns_mean_under_counters"],
            "mfma_pipe_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (simds * cyc),
            "cycles_per_mfma": m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"],
            "other_valu_per_mfma": (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"],
            "lds_insts_per_mfma": m["SQ_INSTS_LDS"] / m["SQ_INSTS_MFMA"],
            "salu_per_mfma": m["SQ_INSTS_SALU"] / m["SQ_INSTS_MFMA"],
            "wave_time_parked_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
            "wave_time_issue_stalled_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            "wave_time_issuing_frac": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"),
            "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
        }
    return out

if __name__ == "__main__":
    root = sys.argv[1]
    res = {"source": "rocprofv3 --pmc, four separate counter passes per kernel (tools/pmc_gemm.sh), eager launches of tools/wq_bench",
           "members": {tag: summarize(f"{root}/pmc_{tag}") for tag in sys.argv[2:]}}
    print(json.dumps(res, indent=1))
