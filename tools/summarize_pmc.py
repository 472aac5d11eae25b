"""Reduce the rocprofv3 --pmc CSVs of tools/profile_round.sh to per-launch HBM traffic of the GEMV.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and counts a wide
coalesced streaming read at exactly half its bytes -> read bytes = FETCH_SIZE * 1024 * 2.
WRITE_SIZE is reported in KiB (uncalibrated; the GEMV writes 8-22 KB per launch, negligible)."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
launches_per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 16     # bench.py: 4 layers x ({q,k,v}, o, {gate,up}, down)
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, eager launches of bench.py's step"}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, f"{tag}_pmc_{ctr}", "*counter_collection.csv"))
    if not files:
        continue
    vals = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != ctr:
                continue
            name = row.get("Kernel_Name", "")
            if "wq_gemv_kernel" not in name and "wq_gemvx_kernel" not in name:
                continue
            key = (row.get("Grid_Size"), row.get("LDS_Block_Size"))
            vals.setdefault(key, []).append(float(row["Counter_Value"]))
    allv = [v for vs in vals.values() for v in vs]
    if allv:
        res[ctr + "_KiB_mean_per_launch"] = sum(allv) / len(allv)
        res[ctr + "_launches"] = len(allv)
        res[ctr + "_by_grid"] = {f"grid={k[0]},lds={k[1]}": sum(v) / len(v) for k, v in vals.items()}
if "FETCH_SIZE_KiB_mean_per_launch" in res:
    rd = res["FETCH_SIZE_KiB_mean_per_launch"] * 1024 * 2
    wr = res.get("WRITE_SIZE_KiB_mean_per_launch", 0.0) * 1024
    res["gemv_hbm_read_bytes_per_launch_corrected"] = rd
    res["gemv_hbm_write_bytes_per_launch"] = wr
    res["gemv_hbm_bytes_per_launch"] = rd + wr
    res["launches_per_step"] = launches_per_step
    res["gemv_hbm_bytes_per_step"] = (rd + wr) * launches_per_step
path = os.path.join(out, f"{tag}_pmc_gemv.json")
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
