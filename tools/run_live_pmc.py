"""bench.py's live HBM-traffic pass alone (rocprofv3 --pmc child runs of the headline step): python tools/run_live_pmc.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = time.time()
print(bench.pmc_traffic_live(16, 4), "seconds", round(time.time() - t, 1))
