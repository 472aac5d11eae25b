"""Seeded random sweep over the operator's configuration space: every configuration the selector
accepts must match the oracle (fp paths 1e-3 relative, integer paths bit exact); configurations it
refuses must be refused loudly at construction - never run and be wrong.

The axes are the MatmulConfig fields of the reference's op tests
(testing/python/operators/test_general_matmul_ops_backend_tl.py:327-343, ..._ops_backend.py:211-229,
..._ops_nf4.py, ..._fp8.py) drawn independently instead of as a hand-picked list, with ragged M and N.
"""
import os

import numpy as np
import pytest

import bitblas_amd as bitblas
from helpers import assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu

W_F16 = ["uint4", "int4", "uint2", "int2", "uint1", "int1", "uint8", "int8", "nf4", "fp4_e2m1", "e4m3_float8"]
W_I8 = ["int4", "uint4", "int2", "uint2", "int1", "int8"]
MS = [1, 2, 3, 5, 7, 8, 13, 16, 17, 33, 64, 100, 128, 200, 257]
NS = [16, 48, 64, 100, 128, 272, 520]
KS = [int(k) for k in os.environ.get("WQAA_SWEEP_KS", "256,512,768,1024,1536,2048").split(",")]


def draw(rng):
    a_int8 = rng.random() < 0.3
    wd = str(rng.choice(W_I8 if a_int8 else W_F16))
    M, N, K = int(rng.choice(MS)), int(rng.choice(NS)), int(rng.choice(KS))
    kw = dict(W_dtype=wd, A_dtype="int8" if a_int8 else "float16")
    if a_int8:
        kw["out_dtype"] = str(rng.choice(["int32", "float32", "float16", "int8"]))   # README.md:79-82
        kw["fast_decoding"] = [None, False, True][int(rng.integers(3))] if wd not in ("int8",) else None
        if wd in ("int4", "uint4"):
            kw["fast_decoding"] = [None, False][int(rng.integers(2))]   # the reference never interleaves int4 for int8
    else:
        kw["out_dtype"] = "float16"
        is_int = wd.startswith(("uint", "int"))
        kw["fast_decoding"] = [None, False, True][int(rng.integers(3))] if is_int and wd not in ("uint8", "int8") else None
        if wd != "fp4_e2m1" and rng.random() < 0.7:
            kw["with_scaling"] = True
            kw["group_size"] = int(rng.choice([-1, 32, 64, 128, 256]))
            kw["scale_mul"] = 0.05
            if is_int and wd.startswith("uint") and rng.random() < 0.6:
                kw["with_zeros"] = True
                kw["zeros_mode"] = str(rng.choice(["original", "rescale", "quantized"]))
        kw["with_bias"] = bool(rng.random() < 0.3)
    return M, N, K, kw


@pytest.mark.parametrize("strict", [None, True], ids=["default_members", "strict_reference"])
@pytest.mark.parametrize("chunk", range(int(os.environ.get("WQAA_SWEEP_CHUNKS", "8"))))   # 40 draws each
def test_random_configurations(chunk, strict):
    """strict=None: the operator as a caller constructs it (the library's default members: at M <= 2 the exact-product GEMV
    family, IEEE e4m3); True: the reference's definition to the letter"""
    rng = np.random.default_rng(1000 + chunk)
    ran = refused = 0
    for _ in range(40):
        M, N, K, kw = draw(rng)
        g = kw.get("group_size", -1)
        if g != -1 and K % g:
            continue
        bit = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[kw["W_dtype"]][1]
        if kw.get("zeros_mode") == "quantized" and (N * bit) % 8:
            continue          # packed zero points need whole bytes per row (QZeros is (K/g, N*bit/8))
        if os.environ.get("WQAA_SWEEP_VERBOSE"):
            print("draw", M, N, K, kw, flush=True)
        try:
            case = make_case(M, N, K, seed=int(rng.integers(1 << 30)), **kw)
            got, mm = hip_output(case, strict_reference=strict)
        except (ValueError, RuntimeError, AssertionError) as e:
            msg = str(e)
            # a refusal must come from the selector / config legalisation, with a reason
            assert any(t in msg for t in ("gemv:", "gemm:", "no gfx950 kernel", "not supported", "Unsupported",
                                          "must", "should be", "scale", "zeros")), (M, N, K, kw, msg)
            refused += 1
            continue
        want = oracle_output(case)
        if kw["A_dtype"] == "int8":
            assert np.array_equal(got, want), (M, N, K, kw, mm.plans[M]["name"])
        else:
            try:
                # (the default members' exact products differ from the reference's per-element rounding by that rounding: DESIGN 4)
                assert_fp_parity(got, want, atol_frac=1e-3 if strict else 1.5e-3)
            except AssertionError as e:
                raise AssertionError(f"{(M, N, K, kw, mm.plans[M]['name'])}: {e}")
        ran += 1
    assert ran >= 15, (ran, refused)


def test_random_bf16_configurations():
    """bfloat16 activations (ref: test_general_matmul_bf16.py; the TE zero modes apply to any A_dtype,
    matmul_dequantize_impl.py:435-449): plain layout, none / scale / the three zeros modes."""
    from test_gemm_gpu import _bf16_case
    rng = np.random.default_rng(77)
    ran = 0
    for _ in range(60):
        wd = str(rng.choice(["uint4", "int4", "uint2", "int2", "uint1", "uint8", "int8", "nf4", "fp4_e2m1", "e4m3_float8"]))
        M, N, K = int(rng.choice(MS)), int(rng.choice([64, 128, 272, 520])), int(rng.choice(KS))
        ws = bool(rng.random() < 0.7)
        g = int(rng.choice([-1, 64, 128, 256])) if ws else -1
        zm = str(rng.choice(["quantized", "original", "rescale"])) if (ws and wd.startswith("uint") and rng.random() < 0.6) else None
        if g != -1 and K % g:
            continue
        try:
            out, want, mm = _bf16_case(M, N, K, wd, g, ws, zm, seed=int(rng.integers(1 << 30)))
        except (ValueError, RuntimeError) as e:
            assert any(t in str(e) for t in ("gemv:", "gemm:", "not supported", "must")), (M, N, K, wd, g, ws, zm, str(e))
            continue
        try:
            assert_fp_parity(out, want, rtol=1e-5, atol_frac=1e-5)
        except AssertionError as e:
            raise AssertionError(f"{(M, N, K, wd, g, ws, zm, mm.plans[M]['name'])}: {e}")
        ran += 1
    assert ran >= 30, ran


def test_random_int4_activation_configurations():
    """packed int4 activations x int4 / int2 weights, every M class and layout, bit exact"""
    from test_int4_act_gpu import run_case
    rng = np.random.default_rng(78)
    for _ in range(40):
        wd = str(rng.choice(["int4", "int2"]))
        fd = False if wd == "int4" else [None, False, True][int(rng.integers(3))]
        M, N, K = int(rng.choice(MS)), int(rng.choice([64, 128, 272, 520])), int(rng.choice([256, 512, 1024, 2048]))
        run_case(M, N, K, wd, fd, seed=int(rng.integers(1 << 30)), out_dtype=str(rng.choice(["int32", "float32"])))


def test_random_groups_equal_single_calls():
    """wqaa_matmul_group over random configurations, member counts and row counts: whether the group fuses into one
    launch (same descriptor apart from N, M <= 2) or runs member by member, every output equals the member's own call
    bit for bit - the definition of the entry (include/wqaa.h)."""
    import torch
    from bitblas_amd import group as wgroup
    from test_group_gpu import build
    from helpers import _to_dev
    rng = np.random.default_rng(4242)
    ran = fused = 0
    for _ in range(60):
        M, N0, K, kw = draw(rng)
        M = int(rng.choice([1, 1, 1, 2, 2, 3, 16]))
        g = kw.get("group_size", -1)
        if g != -1 and K % g:
            continue
        bit = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[kw["W_dtype"]][1]
        count = int(rng.integers(2, 5))
        Ns = [int(rng.choice(NS + [1024, 2048])) for _ in range(count)]
        if kw.get("zeros_mode") == "quantized" and any((n * bit) % 8 for n in Ns):
            continue
        strict = bool(rng.random() < 0.5)
        try:
            cases = [make_case(M, n, K, seed=int(rng.integers(1 << 30)), **kw) for n in Ns]
            for c in cases[1:]:
                c["A"] = cases[0]["A"]
            built = [build(c, strict) for c in cases]
        except (ValueError, RuntimeError, AssertionError):
            continue
        ops, ws = [b[0] for b in built], [b[1] for b in built]
        A = _to_dev(cases[0]["A"], "cuda")
        try:
            single = [op(A, *w) for op, w in zip(ops, ws)]
        except (ValueError, RuntimeError):
            continue            # a configuration the selector refuses (covered by test_random_configurations)
        plan = wgroup.group_plan(ops, M)
        grouped = bitblas.matmul_group(ops, A, ws)
        torch.cuda.synchronize()
        for i, (s, g_) in enumerate(zip(single, grouped)):
            assert torch.equal(s, g_), (M, Ns, K, kw, strict, plan, i)
        ran += 1
        fused += plan["launches"] == 1
    assert ran >= 30 and fused >= 10, (ran, fused)
