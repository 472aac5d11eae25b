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
