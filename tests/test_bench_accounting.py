"""bench.py's accounting, without a GPU: the algorithmic byte counts `roofline.achieved` is computed from are SURVEY.md
section 8(d)'s formula and examples, the PMC traffic figure comes from the newest committed counter file, and the host-core
count respects the container's limits."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_are_the_surveys():
    # c2 (M=1, N=K=4096, g=128, scale only) = 8192 + 8,388,608 + 262,144 + 8192 = 8,667,136 B
    assert bench.algorithmic_bytes(1, 4096, 4096) == 8_667_136
    # with zeros + 262,144 B
    assert bench.algorithmic_bytes(1, 4096, 4096, zeros=True) == 8_667_136 + 262_144
    # c2 (11008 x 4096): 22,544,384 + 704,512 + 8192 + 22,016
    assert bench.algorithmic_bytes(1, 11008, 4096) == 22_544_384 + 704_512 + 8192 + 22_016
    # c4 M=1 (int2 x int8, int32 out, no scale): 4096 + 4,194,304 + 16,384
    assert bench.algorithmic_bytes(1, 4096, 4096, bits=2, scale=False, out_bytes=4, a_bytes=1) == 4096 + 4_194_304 + 16_384
    # the headline step: 4 layers x the 7 Llama-2-7B linears
    per_layer = sum(bench.algorithmic_bytes(1, N, K) for (_, N, K) in bench.LLAMA2_7B_LINEARS)
    assert 4 * per_layer == 418_023_424          # the figure VERDICT r01 recomputed from BENCH_r01.json


def test_pmc_traffic_reads_the_newest_committed_counter_file():
    t = bench.pmc_traffic(16)
    assert t is None or 0.9 * 26_126_464 < t < 1.2 * 26_126_464      # bytes per launch of the 16-launch step (profiles/*pmc*.json)
    newest = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if "pmc" in p and p.endswith(".json"))
    assert newest, "a committed PMC summary is what roofline.traffic quotes"
    with open(os.path.join(ROOT, "profiles", newest[-1])) as f:
        json.load(f)


def test_usable_cores_is_bounded_by_the_affinity_mask():
    n = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_final_line_is_short_enough_for_the_driver_to_parse():
    # BENCH_r04.json.parsed was null because the single stdout line had grown to 24 KB (the driver keeps 8 KB of
    # stdout).  The last line is the contract fields alone, every string clipped; members go to a side file + an earlier line.
    long = "x" * 5000
    result = {
        "metric": long, "value": 3605.123456789, "unit": "GB/s", "n_gpus": 1, "steps": 50, "warmup": 5, "ms_per_step": 0.1159499,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": long, "launches": {str(i): long for i in range(8)}, "launches_per_step": 16,
                   "bytes_per_step_per_gpu": 418_023_424, "sharding": long},
        "roofline": {"bound": "hbm", "achieved": 3605.1234567, "peak": 8000.0, "unit": "GB/s", "frac": 0.45064043, "traffic": 26_289_984.0,
                     "kernel": long, "numerics": long, "bytes_per_launch": 26_126_464.0, "mean_launch_us": 7.2471234, "timing": long,
                     "traffic_source": long},
        "cpu_baseline": {"value": 0.1573, "unit": "GB/s", "cores": 16, "kind": "port", "sample": long},
        "multi_gpu_c5": {"note": long, "tflops": 1234.5678, "ms": 1.25},
    }
    members = {k: {"us_per_launch": 17.123456, "roofline": {"frac": 0.10123456}, "workload": long} for k in bench.MEMBER_KEYS}
    members["broken"] = {"error": long}
    members.update({f"extra{i}": {"us_per_launch": 1.0, "roofline": {"frac": 0.5}} for i in range(80)})
    line = bench.final_line(result, members)
    text = json.dumps(line)
    assert len(text) < 1800, len(text)
    back = json.loads(text)
    assert back["roofline"]["frac"] == 0.4506 and back["roofline"]["bound"] == "hbm" and back["roofline"]["peak"] == 8000.0
    assert back["cpu_baseline"]["value"] == 0.1573 and back["cpu_baseline"]["cores"] == 16 and back["cpu_baseline"]["kind"] == "port"
    assert back["value"] == 3605.1235 and back["n_gpus"] == 1 and back["config"]["launches_per_step"] == 16
    assert set(back["members"]) == set(bench.MEMBER_KEYS) and back["members"]["gemm_uint4_m128"] == {"us": 17.12, "frac": 0.101}
    # the metric's own shapes ride in the contract line, and the line says which numerics the headline is (#8)
    assert {"gemv_int4_n8192k28672", "gemv_int4_n28672k8192", "gemm_uint4_m4096_n28672k8192"} <= set(back["members"])
    assert "strict_reference=0" in back["config"]["numerics"] and "1e-3" in back["config"]["numerics"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    # a failed cpu baseline still gives a line
    result["cpu_baseline"] = {"error": long}
    assert len(json.dumps(bench.final_line(result, None))) < 1800


def test_members_go_to_a_side_file_and_an_earlier_line(tmp_path, capsys):
    members = {"gemm_uint4_m128": {"us_per_launch": 17.1, "roofline": {"frac": 0.1}}, "bad": {"error": "x"}}
    path = tmp_path / "sub" / "members.json"
    bench.emit_members(members, str(path))
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("[bench-members] ")
    assert json.loads(out[0][len("[bench-members] "):])["gemm_uint4_m128"] == {"us": 17.1, "frac": 0.1}
    with open(path) as f:
        rec = json.load(f)
    assert rec["members"] == members and rec["members_summary"]["bad"] == {"error": "x"}
