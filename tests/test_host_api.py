"""Host-side mirror of the reference operator API (no GPU needed).

`MatmulConfig` legalisation and kernel names are checked against vectors produced by RUNNING the
reference's own classes (oracle/gen_config_golden.py -> tests/golden/matmul_config_golden.json):
`repr(config)` is the operator-cache / database key upstream (cache/operator.py:62), so it must match
character for character.  Cache and `Linear` buffer contracts restate testing/python/cache/
test_operator_cache.py and testing/python/module/test_bitblas_linear.py as far as they go without a GPU.
"""
import json
import os
import threading

import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
from bitblas_amd.matmul import MatmulKernelNameGenerator, TransformKind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "matmul_config_golden.json")))


def test_matmul_config_matches_reference_class():
    assert len(GOLDEN) >= 150
    for g in GOLDEN:
        cfg = bitblas.MatmulConfig(**g["kwargs"])
        assert repr(cfg) == g["repr"], g["kwargs"]
        for name, want in g["fields"].items():
            got = getattr(cfg, name)
            got = list(got) if isinstance(got, tuple) else (int(got) if isinstance(got, (TransformKind, bitblas.OptimizeStrategy)) else got)
            assert got == want, (g["kwargs"], name)
        assert MatmulKernelNameGenerator(cfg).generate() == g["kernel_name"]


def test_config_is_hashable_and_frozen():
    a = bitblas.MatmulConfig(M=1, N=256, K=256, W_dtype="int4")
    b = bitblas.MatmulConfig(M=1, N=256, K=256, W_dtype="int4")
    assert a == b and hash(a) == hash(b) and len({a, b}) == 1
    with pytest.raises(Exception):
        a.N = 3
    with pytest.raises(ValueError):
        bitblas.MatmulConfig(M=1, K=16)


def test_operator_construction_plans_without_gpu():
    """Operators are built (selector answers) without a device; forward needs one and says so."""
    cfg = bitblas.MatmulConfig(M=[1, 16, 4096], N=4096, K=4096, A_dtype="float16", W_dtype="uint4", group_size=128,
                               with_scaling=True, with_zeros=True)
    op = bitblas.Matmul(cfg, enable_tuning=False)
    assert op.plans[1]["kernel_family"] == 1 and op.plans[16]["kernel_family"] == 2
    assert op.plans[4096]["block_m"] == 256 and op.plans[4096]["block_n"] == 256   # the 8-wave member
    assert op.retrieve_weight_shape() == [4096, 2048]
    assert op.propagate_a == TransformKind.NonTransform and op.propagate_b == TransformKind.NonTransform
    w = torch.randint(0, 16, (4096, 4096), dtype=torch.int8)
    assert tuple(op.transform_weight(w).shape) == (4096, 2048)
    with pytest.raises(RuntimeError):
        op(torch.zeros(1, 4096, dtype=torch.float16), torch.zeros(4096, 2048, dtype=torch.int8),
           scale=torch.zeros(4096, 32, dtype=torch.float16), zeros=torch.zeros(4096, 32, dtype=torch.float16))


def test_unsupported_config_fails_at_construction():
    with pytest.raises(Exception):
        bitblas.Matmul(bitblas.MatmulConfig(M=1, N=64, K=40, W_dtype="int4"), enable_tuning=False)
    with pytest.raises(ValueError):
        bitblas.Matmul(bitblas.MatmulConfig(M=1, N=64, K=64, layout="nn"), enable_tuning=False)


def test_packed_zero_points_must_fill_whole_bytes():
    """Qzeros is (K / g, N x bits / 8) int8 (tirscript/matmul_dequantize_impl.py:375-389 sizes the row N // 8 * bits): an N that leaves
    a partial byte would read past the row - refused with BAD_DESC instead (round 6)"""
    from bitblas_amd.lib import WqaaError
    kw = dict(K=256, A_dtype="float16", W_dtype="uint2", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="quantized")
    with pytest.raises(WqaaError, match="whole bytes"):
        bitblas.Matmul(bitblas.MatmulConfig(M=1, N=50, **kw), enable_tuning=False)
    bitblas.Matmul(bitblas.MatmulConfig(M=1, N=48, **kw), enable_tuning=False)
    bitblas.Matmul(bitblas.MatmulConfig(M=1, N=50, **dict(kw, zeros_mode="original")), enable_tuning=False)


def test_transform_weight_bytes_are_the_reference_layout():
    import numpy as np
    import wqaa_oracle as oracle
    rng = np.random.default_rng(0)
    w = rng.integers(-8, 8, size=(32, 128)).astype(np.int8)
    for fd in (True, False):
        op = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=32, K=128, W_dtype="int4", fast_decoding=fd), enable_tuning=False)
        got = op.transform_weight(torch.from_numpy(w)).numpy()
        want = oracle.transform_weight(w, "int", 4, fast_decoding=fd)
        assert np.array_equal(got, want)
    # int2 x int8 (BitNet): interleave target width is 8
    w2 = rng.integers(-2, 2, size=(32, 128)).astype(np.int8)
    op = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=32, K=128, A_dtype="int8", W_dtype="int2", accum_dtype="int32",
                                             out_dtype="int32"), enable_tuning=False)
    assert np.array_equal(op.transform_weight(torch.from_numpy(w2)).numpy(),
                          oracle.transform_weight(w2, "int", 2, fast_decoding=True, a_dtype="int8"))


def test_operator_cache_roundtrip(tmp_path):
    """cache/test_operator_cache.py: add/get/size, save_into_database + load_from_database."""
    cache = bitblas.OperatorCache()
    cfg = bitblas.MatmulConfig(M=1, N=256, K=256, W_dtype="uint4", with_scaling=True, group_size=128)
    assert cache.get(cfg) is None and cache.size() == 0
    op = bitblas.Matmul(cfg, enable_tuning=False)
    cache.add(cfg, op)
    assert cache.get(bitblas.MatmulConfig(M=1, N=256, K=256, W_dtype="uint4", with_scaling=True, group_size=128)) is op
    cache.save_into_database(str(tmp_path), target="hip -mcpu=gfx950")
    fresh = bitblas.OperatorCache()
    fresh.load_from_database(str(tmp_path), target="hip -mcpu=gfx950")
    assert fresh.size() == 1 and fresh.get(cfg) is not None
    assert repr(fresh.get(cfg).config) == repr(cfg)


def test_operator_cache_concurrent_creation():
    """cache/test_operator_cache_spin_lock.py: several threads asking for the same operator."""
    cache = bitblas.OperatorCache()
    cfg = bitblas.MatmulConfig(M=1, N=128, K=128, W_dtype="int4")
    made = []

    def worker():
        with cache.cache_locker:
            op = cache.get(cfg)
            if op is None:
                op = bitblas.Matmul(cfg, enable_tuning=False)
                cache.add(cfg, op)
                made.append(1)

    threads = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert len(made) == 1 and cache.size() == 1


def test_linear_buffers_and_validation():
    """module/test_bitblas_linear.py contracts that do not need a device: buffer names/shapes/dtypes,
    state_dict round trip, argument validation (module/__init__.py:155-205)."""
    lin = bitblas.Linear(1024, 512, bias=True, A_dtype="float16", W_dtype="uint4", group_size=128,
                         with_scaling=True, with_zeros=True, zeros_mode="quantized", opt_M=[1, 16], enable_tuning=False)
    sd = lin.state_dict()
    assert tuple(sd["qweight"].shape) == (512, 512) and sd["qweight"].dtype == torch.int8
    assert tuple(sd["scales"].shape) == (512, 8) and sd["scales"].dtype == torch.float16
    assert tuple(sd["zeros"].shape) == (8, 512 // 8 * 4) and sd["zeros"].dtype == torch.int8
    assert tuple(sd["bias"].shape) == (512,)
    lin2 = bitblas.Linear(1024, 512, bias=True, A_dtype="float16", W_dtype="uint4", group_size=128,
                          with_scaling=True, with_zeros=True, zeros_mode="quantized", opt_M=[1, 16], enable_tuning=False)
    lin2.load_state_dict(sd)
    assert lin2.bitblas_matmul is lin.bitblas_matmul      # global operator cache hit
    with pytest.raises(ValueError):
        bitblas.Linear(100, 512, W_dtype="uint4")
    with pytest.raises(ValueError):
        bitblas.Linear(1024, 512, W_dtype="uint4", group_size=100)
    dense = bitblas.Linear(256, 128, A_dtype="float16", W_dtype="float16", opt_M=1, enable_tuning=False)
    assert tuple(dense.weight.shape) == (128, 256)


def test_public_names():
    for name in ("Matmul", "MatmulConfig", "MatmulWithSplitK", "MatmulConfigWithSplitK", "Linear", "set_log_level",
                 "auto_detect_nvidia_target", "global_operator_cache", "general_compress", "interleave_weight"):
        assert hasattr(bitblas, name)
    assert bitblas.auto_detect_nvidia_target().startswith("hip")
    bitblas.set_log_level("DEBUG")
    bitblas.set_log_level("WARNING")


def test_split_k_operator_mirror():
    """`MatmulWithSplitK` (ops/general_matmul_splitk.py): same construction surface, k_split kept as a hint"""
    cfg = bitblas.MatmulConfigWithSplitK(M=16, N=1024, K=4096, A_dtype="float16", W_dtype="int4", k_split=4)
    assert cfg.k_split == 4 and "k_split=4" in repr(cfg) and cfg.fast_decoding is True
    op = bitblas.MatmulWithSplitK(cfg, enable_tuning=False)
    assert op.k_split == 4 and op.plans[16]["kernel_family"] == 2
    assert op.retrieve_weight_shape() == [1024, 2048]


def test_k_split_reaches_the_selector():
    """`MatmulConfigWithSplitK.k_split` (ops/general_matmul_splitk.py:21-23) travels as `wqaa_matmul_desc.k_split_hint`:
    it sets the split of the members whose K split is a free parameter and is reported back in the plan"""
    kw = dict(N=4096, K=4096, A_dtype="float16", W_dtype="int4", group_size=128, with_scaling=True)
    # pipelined MFMA member (M = 128): the split-K count
    for ks in (2, 8):
        op = bitblas.MatmulWithSplitK(bitblas.MatmulConfigWithSplitK(M=128, k_split=ks, **kw), enable_tuning=False)
        assert op.plans[128]["split_k"] == ks, op.plans[128]
    # clamped to the k-steps there are (K = 512: four 128-deep steps) and to 16
    op = bitblas.MatmulWithSplitK(bitblas.MatmulConfigWithSplitK(M=128, N=4096, K=512, A_dtype="float16", W_dtype="int4",
                                                                 group_size=128, with_scaling=True, k_split=64), enable_tuning=False)
    assert op.plans[128]["split_k"] == 4
    # the plain operator decides for itself (round 5: the mid-M member - K in 8 slices, its own second launch adds them)
    plain = bitblas.Matmul(bitblas.MatmulConfig(M=128, **kw), enable_tuning=False).plans[128]
    assert plain["split_k"] == 8 and plain["name"].endswith("xmk"), plain
    # M = 1 exact-product GEMV: the K split across the waves of a workgroup
    kw.update(N=1024, K=16384)          # four 4096-deep steps of the 4-bit GEMV
    op = bitblas.MatmulWithSplitK(bitblas.MatmulConfigWithSplitK(M=1, k_split=2, **kw), enable_tuning=False, strict_reference=False)
    assert op.plans[1]["split_k"] == 2 and op.plans[1]["name"].endswith("k2"), op.plans[1]
    # structural splits (one-launch decode member at M = 16) keep their own
    op = bitblas.MatmulWithSplitK(bitblas.MatmulConfigWithSplitK(M=16, k_split=4, **kw), enable_tuning=False)
    assert op.plans[16]["name"].endswith("xdl")


@pytest.mark.parametrize("bits", [4, 2])
def test_gptq_repack_against_reference_run_vectors(bits):
    """tests/golden/gptq_golden.npz: what the reference's own unpack_qweight / unpack_qzeros[_v2] and
    Linear.repack_from_gptq[_v2] (bitblas/module/__init__.py:24-74, 315-363) produce on seeded AutoGPTQ-shaped
    tensors (oracle/gen_gptq_golden.py runs them).  The mirror must produce the same buffers - including the
    int8 wrap of `stored zero + 1` at the top of the code range."""
    import types
    import wqaa_oracle as oracle
    from bitblas_amd import lib as wlib
    from bitblas_amd import module as wmod
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gptq_golden.npz"))
    tag = f"b{bits}"
    qweight, qzeros, scales = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("qweight", "qzeros", "scales"))
    K, N = qweight.shape[0] * 32 // bits, qweight.shape[1]
    gsz = K // qzeros.shape[0]
    codes_ref = g[f"{tag}_unpack_qweight"]                       # (N, K) integer codes
    # helpers: product mirror and oracle
    assert np.array_equal(wmod.unpack_qweight(qweight.T.contiguous().view(torch.int8), bits).numpy(), codes_ref)
    assert np.array_equal(wmod.unpack_qzeros(qzeros, bits).numpy(), g[f"{tag}_unpack_qzeros"])
    assert np.array_equal(wmod.unpack_qzeros_v2(qzeros, bits).numpy(), g[f"{tag}_unpack_qzeros_v2"])
    assert np.array_equal(oracle.unpack_qweight(qweight.T.contiguous().view(torch.int8).numpy(), bits), codes_ref)
    assert np.array_equal(oracle.unpack_qzeros(qzeros.numpy(), bits), g[f"{tag}_unpack_qzeros"])
    assert np.array_equal(oracle.unpack_qzeros(qzeros.numpy(), bits, v2=True), g[f"{tag}_unpack_qzeros_v2"])
    fake = types.SimpleNamespace(qweight=qweight, qzeros=qzeros, scales=scales, bias=None)
    for v2 in (False, True):
        for mode in ("original", "rescale", "quantized"):
            lin = bitblas.Linear(K, N, bias=False, A_dtype="float16", W_dtype=f"uint{bits}", accum_dtype="float16",
                                 out_dtype="float16", group_size=gsz, with_scaling=True, with_zeros=True,
                                 zeros_mode=mode, opt_M=[1, 16])
            (lin.repack_from_gptq_v2 if v2 else lin.repack_from_gptq)(fake, device="cpu")
            key = f"{tag}_{'v2' if v2 else 'v1'}_{mode}"
            assert np.array_equal(lin.scales.numpy(), g[f"{key}_scales"])
            assert np.array_equal(lin.zeros.numpy(), g[f"{key}_zeros"]), key
            cfg = lin.bitblas_matmul.config
            layout = wlib.LAYOUT_LOP3 if cfg.fast_decoding else wlib.LAYOUT_PLAIN
            back = wlib.unpack_weight(lin.qweight.numpy(), K, bits, layout, wlib.F16)
            assert np.array_equal(back, codes_ref)


def test_bench_algorithmic_bytes_are_the_survey_figures():
    """bench.py prices `roofline.achieved` with SURVEY.md section 8(d)'s per-launch byte counts."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.algorithmic_bytes(1, 4096, 4096) == 8_667_136                      # c2, scale only
    assert bench.algorithmic_bytes(1, 4096, 4096, zeros=True) == 8_667_136 + 262_144
    assert bench.algorithmic_bytes(1, 11008, 4096) == 8192 + 22_544_384 + 704_512 + 22_016
    assert bench.algorithmic_bytes(1, 1024, 1024, g=1024) == 2048 + 524_288 + 2048 + 2048          # c1, g = -1
    names = [n for (n, _, _) in bench.LLAMA2_7B_LINEARS]
    assert len(names) == 7 and bench.HBM_PEAK_GBS == 8000.0


def test_operator_cache_persists_the_tuned_threshold(tmp_path):
    """what hardware_aware_finetune measured (desc.two_pass_min_m) travels with the operator database
    (reference: cache/operator.py:62-120 stores the tuned source next to the config)"""
    import bitblas_amd as bb
    from bitblas_amd.cache import OperatorCache
    cfg = bb.MatmulConfig(M=[1, 4096], N=1024, K=1024, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True,
                          with_zeros=True)
    op = bb.Matmul(cfg, enable_tuning=False)
    assert op._desc.two_pass_min_m == 0
    op._desc.two_pass_min_m = 1024
    c = OperatorCache()
    c.add(cfg, op)
    d = c.save_into_database(str(tmp_path), target="hip")
    c2 = OperatorCache()
    c2.load_from_database(d, target="hip")
    assert c2.size() == 1 and c2.get(cfg)._desc.two_pass_min_m == 1024


def test_shared_workspace_grows_geometrically_and_keeps_retired_buffers():
    """`lib.shared_workspace`: ONE large scratch per (stream, device) for all operators (the two-pass member's B_decode); a
    buffer that is replaced stays alive - a captured hipGraph may still point at it (the C pool's rule, csrc/wqaa_gemm.hip)"""
    from bitblas_amd import lib as wlib
    wlib._shared_ws.clear()
    del wlib._shared_ws_retired[:]
    a = wlib.shared_workspace(7, "cpu", 1000)
    assert a.numel() >= 1000 and wlib.shared_workspace(7, "cpu", 500) is a          # reused while it is large enough
    b = wlib.shared_workspace(7, "cpu", 1500)
    assert b is not a and b.numel() >= 2 * a.numel() and wlib._shared_ws_retired == [a]
    c = wlib.shared_workspace(8, "cpu", 10)                                          # another stream: its own buffer
    assert c is not b and wlib.shared_workspace(7, "cpu", 1) is b
    wlib._shared_ws.clear()
    del wlib._shared_ws_retired[:]


def test_operator_workspace_is_retired_not_freed_when_a_taller_batch_needs_more(monkeypatch):
    """`BoundLib.run`: the per-(operator, stream) split-K scratch of a dynamic-M operator grows with the row count; the
    buffer it replaces must stay alive - a hipGraph captured at the smaller row count still holds its address"""
    from bitblas_amd import lib as wlib
    del wlib._shared_ws_retired[:]
    desc = wlib.make_desc(N=256, K=1024, a_dtype=wlib.F16, w_format=wlib.W_UINT, w_bits=4, out_dtype=wlib.F16, group_size=128,
                          with_scaling=True)
    b = wlib.BoundLib(desc, has_lut=False, dynamic_m=True)
    seen = []
    monkeypatch.setattr(b, "workspace_need", lambda m: {48: 4096, 64: 6000, 32: 1000}[m])
    monkeypatch.setattr(b, "run_ws", lambda *a: seen.append((a[-2], a[-1])))
    b.run(1, 2, None, 3, None, None, 4, 48, 0, "cpu")
    first = b._ws[(0, "cpu")]
    b.run(1, 2, None, 3, None, None, 4, 32, 0, "cpu")                 # a shorter batch reuses it
    assert b._ws[(0, "cpu")] is first and seen[1] == (first.data_ptr(), 1000)
    b.run(1, 2, None, 3, None, None, 4, 64, 0, "cpu")                 # a taller one outgrows it
    second = b._ws[(0, "cpu")]
    assert second is not first and second.numel() >= 2 * first.numel()
    assert wlib._shared_ws_retired == [first] and seen[2] == (second.data_ptr(), 6000)
    del wlib._shared_ws_retired[:]


@pytest.mark.parametrize("bits", [1, 2, 4])
def test_packer_word_paths_match_the_definition(bits):
    """csrc/wqaa_pack.hip gathers a word's fields with 64-bit SWAR shifts and changes layout through byte tables; the definition is
    the reference's `general_compress` + `interleave_weight` (run as the oracle's restatement, itself pinned by
    tests/golden/packing_golden.npz): whole words, row tails, one row and many (threads), source bytes with garbage above the field"""
    import ctypes
    import wqaa_oracle as oracle
    from bitblas_amd import lib as wlib
    L = wlib.load_library()
    for f in (L.wqaa_pack_weight, L.wqaa_unpack_weight):
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.wqaa_relayout_weight.restype = ctypes.c_int
    L.wqaa_relayout_weight.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    rng = np.random.default_rng(bits)
    mask = (1 << bits) - 1
    for cols in (32, 96, 40, 24, 200, 8192):
        if cols % (8 // bits):
            continue
        for rows in (1, 5, 700):
            src = rng.integers(-128, 128, size=(rows, cols)).astype(np.int8)
            fields = (src.astype(np.int16) & mask).astype(np.int8)
            plain = oracle.general_compress(fields, bits)
            out = np.empty((rows, cols * bits // 8), dtype=np.int8)
            assert L.wqaa_pack_weight(src.ctypes.data, rows, cols, bits, wlib.LAYOUT_PLAIN, wlib.DTYPE_CODE["float16"], out.ctypes.data) == wlib.OK
            assert np.array_equal(out.view(np.uint8), plain.view(np.uint8))
            if (cols * bits // 8) % 4:
                continue
            for target in ("float16", "int8"):
                want = np.asarray(oracle.interleave_weight(plain.copy(), bits, target)).view(np.uint8).reshape(out.shape)
                lop3 = np.empty_like(out)
                assert L.wqaa_pack_weight(src.ctypes.data, rows, cols, bits, wlib.LAYOUT_LOP3, wlib.DTYPE_CODE[target], lop3.ctypes.data) == wlib.OK
                assert np.array_equal(lop3.view(np.uint8), want)
                back = np.empty_like(src)
                assert L.wqaa_unpack_weight(lop3.ctypes.data, rows, cols, bits, wlib.LAYOUT_LOP3, wlib.DTYPE_CODE[target], back.ctypes.data) == wlib.OK
                assert np.array_equal(back, fields)
                moved = np.empty_like(out)
                assert L.wqaa_relayout_weight(out.ctypes.data, rows, out.shape[1], bits, wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3, wlib.DTYPE_CODE[target],
                                              moved.ctypes.data) == wlib.OK
                assert np.array_equal(moved.view(np.uint8), want)
