"""Split-K scratch ownership (VERDICT r01 weak #6 / ADVICE r01): the partial sums of the split-K MFMA members live in
a slab per (device, stream) that is retired - never freed - when it has to grow, or in a caller-owned workspace
(`wqaa_matmul_opts`), the reference's ownership model (bitblas/ops/general_matmul/__init__.py:29, 456-457, 482).

  * two streams running split-K members concurrently do not see each other's partial sums;
  * a hipGraph captured before the slab grew replays correctly afterwards;
  * growth during stream capture is refused loudly, a caller-owned workspace works inside capture;
  * `wqaa_workspace_bytes` reports what the selected member needs.
"""
import numpy as np
import pytest
import torch

from helpers import set_knobs, assert_fp_parity, make_case, oracle_output

import bitblas_amd as bitblas
from bitblas_amd import lib as wl

pytestmark = pytest.mark.gpu


def _prepared(M, N, K, seed):
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.05, seed=seed)
    mm = bitblas.Matmul(case["config"], enable_tuning=False)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()        # noqa: E731
    ops = dict(A=dev(case["A"]), W=mm.transform_weight(torch.from_numpy(case["w_user"]).cuda()), scale=dev(case["scale"]),
               zeros=dev(case["zeros"]))
    return case, mm, ops


def _run(mm, ops, out, stream, m):
    """the plain C entry (wqaa_matmul): scratch from the LIBRARY's per-(device, stream) pool"""
    import ctypes
    lib = wl.load_library()
    st = lib.wqaa_matmul(ctypes.byref(mm.lib.desc), ops["A"].data_ptr(), ops["W"].data_ptr(), None, ops["scale"].data_ptr(),
                         ops["zeros"].data_ptr(), None, out.data_ptr(), m, stream.cuda_stream)
    wl.check(st)


def test_python_operator_owns_its_workspace_and_captures_on_a_fresh_stream():
    """`Matmul.forward` / `lib.run`: caller-owned workspace from torch's allocator, one per (operator, stream) -
    `torch.cuda.graph` captures on a side stream no call has run on before, and that has to work"""
    case, mm, ops = _prepared(48, 1024, 8192, 6)
    assert mm.lib.workspace_bytes(48) > 0
    A, W = ops["A"], ops["W"]
    g = torch.cuda.CUDAGraph()
    out = torch.empty((48, 1024), dtype=torch.float16, device="cuda")
    with torch.cuda.graph(g):
        mm(A, W, scale=ops["scale"], zeros=ops["zeros"], output=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    want = oracle_output(case)
    assert_fp_parity(out.cpu().numpy(), want)
    # two streams through the Python operator: each gets its own workspace tensor
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1, o2 = torch.empty_like(out), torch.empty_like(out)
    for _ in range(16):
        with torch.cuda.stream(s1):
            mm(A, W, scale=ops["scale"], zeros=ops["zeros"], output=o1)
        with torch.cuda.stream(s2):
            mm(A, W, scale=ops["scale"], zeros=ops["zeros"], output=o2)
    torch.cuda.synchronize()
    assert_fp_parity(o1.cpu().numpy(), want)
    assert torch.equal(o1, o2)
    assert len(mm.lib._ws) >= 3


def test_workspace_bytes_query():
    _, mm, _ = _prepared(32, 512, 4096, 0)
    plan = mm.plans[32]
    assert plan["split_k"] > 1, plan
    assert mm.lib.workspace_bytes(32) == plan["split_k"] * 32 * 512 * 4
    gemv = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=512, K=4096, A_dtype="float16", W_dtype="int4", group_size=128,
                                               with_scaling=True), enable_tuning=False)
    assert gemv.lib.workspace_bytes(1) == 0


def test_two_streams_do_not_share_partial_sums():
    """the same split-K member on two streams at once, different activations: with one shared scratch buffer the
    reduce launch of one stream sums partials the other stream is overwriting"""
    M, N, K = 32, 512, 4096
    case_a, mm, ops_a = _prepared(M, N, K, 1)
    case_b = dict(case_a)
    rng = np.random.default_rng(7)
    case_b["A"] = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
    ops_b = dict(ops_a, A=torch.from_numpy(case_b["A"]).cuda())
    assert mm.plans[M]["split_k"] > 1
    want_a, want_b = oracle_output(case_a), oracle_output(case_b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a = [torch.empty((M, N), dtype=torch.float16, device="cuda") for _ in range(64)]
    outs_b = [torch.empty((M, N), dtype=torch.float16, device="cuda") for _ in range(64)]
    torch.cuda.synchronize()
    for oa, ob in zip(outs_a, outs_b):
        _run(mm, ops_a, oa, s1, M)
        _run(mm, ops_b, ob, s2, M)
    torch.cuda.synchronize()
    for oa, ob in zip(outs_a, outs_b):
        assert_fp_parity(oa.cpu().numpy(), want_a)
        assert_fp_parity(ob.cpu().numpy(), want_b)
    # bit-stable as well: every repetition gives the same bits (fixed summation order, private scratch)
    assert all(torch.equal(outs_a[0], o) for o in outs_a[1:])
    assert all(torch.equal(outs_b[0], o) for o in outs_b[1:])


def test_graph_captured_before_the_scratch_grew_still_replays():
    small_case, mm_s, ops_s = _prepared(24, 256, 4096, 2)   # M = 24: the skinny split-K member (M <= 16 is the one-launch decode member)
    big_case, mm_b, ops_b = _prepared(64, 8192, 8192, 3)
    assert mm_s.plans[24]["split_k"] > 1, mm_s.plans[24]
    s = torch.cuda.Stream()
    out_s = torch.empty((24, 256), dtype=torch.float16, device="cuda")
    out_b = torch.empty((64, 8192), dtype=torch.float16, device="cuda")
    with torch.cuda.stream(s):
        _run(mm_s, ops_s, out_s, s, 24)          # first call outside capture: the stream's slab exists
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            _run(mm_s, ops_s, out_s, s, 24)
        g.replay()
        s.synchronize()
        want_s = oracle_output(small_case)
        assert_fp_parity(out_s.cpu().numpy(), want_s)
        need_big = mm_b.lib.workspace_bytes(64)
        assert need_big > (8 << 20), "the big shape must outgrow the initial slab for this test to mean anything"
        _run(mm_b, ops_b, out_b, s, 64)          # grows the slab of stream s: the old one is retired, not freed
        s.synchronize()
        out_s.zero_()
        for _ in range(8):
            g.replay()                             # writes partials into the retired slab: still allocated
        s.synchronize()
    assert_fp_parity(out_s.cpu().numpy(), want_s)
    rows = np.arange(0, 64, 7)
    sub = dict(big_case)
    sub["A"] = big_case["A"][rows]
    assert_fp_parity(out_b.cpu().numpy()[rows], oracle_output(sub))


def test_growth_during_capture_is_refused_and_a_caller_workspace_works():
    case, mm, ops = _prepared(48, 1024, 8192, 4)
    need = mm.lib.workspace_bytes(48)
    assert need > 0
    s = torch.cuda.Stream()                        # a fresh stream has no slab yet
    out = torch.empty((48, 1024), dtype=torch.float16, device="cuda")
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with pytest.raises(wl.WqaaError, match="capture"):
            with torch.cuda.graph(g, stream=s):
                _run(mm, ops, out, s, 48)
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            mm.lib.run_ws(ops["A"].data_ptr(), ops["W"].data_ptr(), None, ops["scale"].data_ptr(), ops["zeros"].data_ptr(),
                          None, out.data_ptr(), 48, s.cuda_stream, ws.data_ptr(), need)
        g2.replay()
        s.synchronize()
    assert_fp_parity(out.cpu().numpy(), oracle_output(case))
    with pytest.raises(wl.WqaaError, match="workspace"):
        mm.lib.run_ws(ops["A"].data_ptr(), ops["W"].data_ptr(), None, ops["scale"].data_ptr(), ops["zeros"].data_ptr(),
                      None, out.data_ptr(), 48, torch.cuda.current_stream().cuda_stream, ws.data_ptr(), 1 << 16)
    # (round 5: `workspace_bytes` is the larger of the mid-M member's exchange buffer and the split-K member's partial sums it falls back
    # to; a workspace that holds either is served - one that holds neither is refused)


def test_automatic_two_pass_does_not_refuse_a_capture_that_needed_no_scratch_before():
    """round 4: float16 x int8 at >= 1024 rows runs B_decode + the dense member - which needs N K 2 bytes of scratch.  Through the
    plain C entry on a fresh stream under capture (no caller workspace, no earlier call) the library's pool cannot grow: the call
    must take the fused member instead of failing, and the same call outside capture afterwards takes the two-pass form; both
    meet the oracle on sampled rows"""
    import ctypes
    import wqaa_oracle as oracle
    M, N, K = 2048, 2048, 1024
    rng = np.random.default_rng(5)
    A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
    W = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="int8", accum_dtype="float32", out_dtype="float16"),
                        enable_tuning=False)
    assert "_dq_" in mm.plans[M]["name"] and mm.lib.workspace_bytes(M) >= N * K * 2, mm.plans[M]
    Ad, Wd = torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda()
    lib = wl.load_library()
    rows = np.arange(0, M, 61)
    want = oracle.matmul_dequant(A[rows], W, source_format="int", bit=8, a_dtype="float16", out_dtype="float32")

    def call(out, stream):
        wl.check(lib.wqaa_matmul(ctypes.byref(mm.lib.desc), Ad.data_ptr(), Wd.data_ptr(), None, None, None, None, out.data_ptr(), M, stream.cuda_stream))

    s = torch.cuda.Stream()                        # a fresh stream: no slab of the library's pool yet
    out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            call(out, s)                           # must not raise
        g.replay()
        s.synchronize()
        assert_fp_parity(out.float().cpu().numpy()[rows], want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-3)
        out2 = torch.zeros_like(out)
        call(out2, s)                              # outside capture: the pool grows, B_decode + dense
        s.synchronize()
    assert_fp_parity(out2.float().cpu().numpy()[rows], want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-3)
    # (the fused member of this shape splits K: another summation order - the two forms agree to rounding, not bit for bit)
    assert_fp_parity(out.float().cpu().numpy()[rows], out2.float().cpu().numpy()[rows], rtol=1e-3, atol_frac=1e-3)


def test_automatic_two_pass_is_capped_and_takes_a_short_workspace_as_the_fused_member(monkeypatch):
    """round 5: the automatic two-pass form must not turn a call that needed no scratch into a refused or a memory-hungry
    one.  Beyond WQAA_TWO_PASS=auto_max_mb=N the plan, `workspace_bytes` and the call all say "fused member"; with the cap open, a caller
    workspace too small for B_decode runs the fused member instead of returning BAD_DESC; both meet the oracle."""
    import ctypes
    import wqaa_oracle as oracle
    M, N, K = 2048, 2048, 1024
    rng = np.random.default_rng(6)
    A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
    W = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="int8", accum_dtype="float32", out_dtype="float16")
    rows = np.arange(0, M, 67)
    want = oracle.matmul_dequant(A[rows], W, source_format="int", bit=8, a_dtype="float16", out_dtype="float32").astype(np.float16).astype(np.float32)
    Ad, Wd = torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda()
    # (a) the cap: N K 2 = 4 MiB of scratch against a 1 MiB cap
    set_knobs(monkeypatch, "two_pass", auto_max_mb="1")
    capped = bitblas.Matmul(cfg, enable_tuning=False)
    assert "_dq_" not in capped.plans[M]["name"], capped.plans[M]["name"]
    fused_need = capped.lib.workspace_bytes(M)
    assert fused_need < N * K * 2
    out = capped(Ad, Wd)
    torch.cuda.synchronize()
    assert_fp_parity(out.float().cpu().numpy()[rows], want, rtol=1e-3, atol_frac=1e-3)
    # (b) cap open, the caller's workspace holds the fused member's partial sums but not B_decode
    set_knobs(monkeypatch, "two_pass", auto_max_mb=None)
    mm = bitblas.Matmul(cfg, enable_tuning=False)
    need = mm.lib.workspace_bytes(M)
    assert "_dq_" in mm.plans[M]["name"] and need >= N * K * 2
    small = torch.empty(max(fused_need, 4096), dtype=torch.uint8, device="cuda")      # what the fused member needs - not B_decode's 4 MiB
    out2 = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    mm.lib.run_ws(Ad.data_ptr(), Wd.data_ptr(), None, None, None, None, out2.data_ptr(), M, torch.cuda.current_stream().cuda_stream, small.data_ptr(), small.numel())
    torch.cuda.synchronize()
    assert_fp_parity(out2.float().cpu().numpy()[rows], want, rtol=1e-3, atol_frac=1e-3)
