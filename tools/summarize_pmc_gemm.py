"""gpurun_out/pmc_<tag>/ (tools/pmc_gemm.sh) -> one JSON with the per-launch counter means of the GEMM kernel
and the figures derived from them (matrix-pipe utilisation, instruction mix per MFMA, wave-state split)."""
import csv, glob, json, sys, collections

def summarize(d):
    """the library kernel with the largest total duration in the run (wq_gemm_kernel, wq_gemm_decode_kernel or
    wq_gemv_kernel - the split-K reduce launch is never the one of interest)"""
    rows = []
    for f in glob.glob(d + "/*counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f))
                 if "wqaa::wq_" in r["Kernel_Name"] and "splitk_reduce" not in r["Kernel_Name"]]
    total = collections.Counter()
    for r in rows:
        total[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = total.most_common(1)[0][0] if total else None
    acc = collections.defaultdict(list)
    durs = []
    for r in rows:
        if r["Kernel_Name"] == name:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    out = {"kernel": name, "counters_mean_per_launch": m, "launch_ns_mean_under_counters": sum(durs) / max(1, len(durs))}
    if "GRBM_GUI_ACTIVE" in m and not m.get("SQ_INSTS_MFMA"):
        # GEMV-family kernel (no MFMA): wave-state split and memory-pipe figures only
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        out["derived"] = {
            "cycles_per_launch": cyc,
            "clock_GHz": cyc / out["launch_ns_mean_under_counters"],
            "valu_per_vmem_read": m["SQ_INSTS_VALU"] / m["SQ_INSTS_VMEM_RD"] if m.get("SQ_INSTS_VMEM_RD") else None,
            "lds_insts_per_vmem_read": m["SQ_INSTS_LDS"] / m["SQ_INSTS_VMEM_RD"] if m.get("SQ_INSTS_VMEM_RD") and "SQ_INSTS_LDS" in m else None,
            "wave_time_parked_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
            "wave_time_issue_stalled_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            "wave_time_issuing_frac": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"] if "SQ_ACTIVE_INST_ANY" in m else None,
            "mean_waves_resident": m["SQ_WAVE_CYCLES"] / (m["SQ_BUSY_CYCLES"] / 8 * 4) if m.get("SQ_BUSY_CYCLES") else None,
            "ta_busy_frac": m["TA_BUSY_avr"] / cyc if "TA_BUSY_avr" in m else None,
            "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
        }
    elif "SQ_INSTS_MFMA" in m and "GRBM_GUI_ACTIVE" in m:
        simds, xcds = 1024, 8
        cyc = m["GRBM_GUI_ACTIVE"] / xcds                      # busy cycles of one XCD
        # GRBM_GUI_ACTIVE keeps counting past a short kernel's own timestamps (ramp-up / drain of the whole dispatch): round 5's
        # mid-M row came out at an impossible 3.49 GHz.  Where the quotient exceeds the chip's 2.4 GHz, the kernel's own wall time
        # at 2.4 GHz bounds its cycles and no clock is reported
        wall_cyc = out["launch_ns_mean_under_counters"] * 2.4
        clock = cyc / out["launch_ns_mean_under_counters"]
        if clock > 2.45:
            cyc, clock = wall_cyc, None
        out["derived"] = {
            "cycles_per_launch": cyc,
            "clock_GHz": clock,
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
