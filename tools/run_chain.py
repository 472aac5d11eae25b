"""bench.py's `step_chained` member alone, three repeats: the decode step chained through its data with the layer's RMSNorms,
residual adds and gated activation, fused (4 launches per layer) against composed (torch's kernels between the projections).
    python tools/run_chain.py        # one JSON line per repeat (profiles/r03_step_chained*.txt)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
for i in range(3):
    r = bench.time_step_chained(dev, gen)
    print(json.dumps({k: r[k] for k in ("fused", "composed", "fused_vs_composed_max_rel_err", "bit_identical")}))
