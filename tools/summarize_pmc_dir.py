"""tools/summarize_pmc_dir.py <dir> [launches_per_step]: per-launch HBM traffic of the GEMV from <dir>/pmc_FETCH_SIZE and
<dir>/pmc_WRITE_SIZE (rocprofv3 --pmc passes of bench.py's step, tools/r04_final.sh) - tools/summarize_pmc.py's reduction for
the per-round directory layout.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and counts a wide
coalesced streaming read at half its bytes -> read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE in KiB."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
lps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, eager launches of bench.py's step"}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(d, f"pmc_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for fn in files:
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == ctr and ("wq_gemv_kernel" in row.get("Kernel_Name", "") or "wq_gemvx_kernel" in row.get("Kernel_Name", "")):
                    vals.append(float(row["Counter_Value"]))
    if vals:
        res[ctr + "_KiB_mean_per_launch"] = sum(vals) / len(vals)
        res[ctr + "_launches"] = len(vals)
if "FETCH_SIZE_KiB_mean_per_launch" in res:
    rd = res["FETCH_SIZE_KiB_mean_per_launch"] * 1024 * 2
    wr = res.get("WRITE_SIZE_KiB_mean_per_launch", 0.0) * 1024
    res.update({"gemv_hbm_read_bytes_per_launch_corrected": rd, "gemv_hbm_write_bytes_per_launch": wr, "gemv_hbm_bytes_per_launch": rd + wr,
                "launches_per_step": lps, "gemv_hbm_bytes_per_step": (rd + wr) * lps})
print(json.dumps(res, indent=1))
