"""Shared case builder for the parity tests: seeded inputs, the HIP path through the C ABI
(`bitblas_amd.Matmul` -> libwqaa_hip.so) and the CPU oracle on the same data.

Input recipe = the reference's own op test (testing/python/operators/
test_general_matmul_ops_backend_tl.py:170-218): A = rand - 0.5, integer codes uniform over the code
range, scale = rand, zeros = 2^(bit-1) in the three zero modes.
"""
from __future__ import annotations

import os

import numpy as np
import torch

import wqaa_oracle as oracle

import bitblas_amd as bitblas


def make_case(M, N, K, W_dtype="int4", A_dtype="float16", out_dtype="float16", group_size=-1,
              with_scaling=False, with_zeros=False, zeros_mode="original", with_bias=False,
              fast_decoding=None, seed=0, scale_mul=1.0, accum_dtype=None):
    rng = np.random.default_rng(seed)
    source_format, bit = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[W_dtype]
    g = K if group_size == -1 else group_size
    case = dict(M=M, N=N, K=K, source_format=source_format, bit=bit, g=g)
    if A_dtype == "float16":
        A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
    elif A_dtype == "int8":
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    else:
        raise NotImplementedError(A_dtype)
    case["A"] = A
    # integer weights as the user would hold them (signed for int formats), and storage codes
    big = N * K > (1 << 26)      # the Llama-70B linears: draw int8 directly (an int64 draw of 235 M elements takes 10 s)
    if source_format == "uint":
        if big and bit < 8:
            w_user = rng.integers(0, 1 << bit, size=(N, K), dtype=np.int8)
        else:
            w_user = rng.integers(0, 1 << bit, size=(N, K)).astype(np.int8) if bit < 8 else \
                rng.integers(0, 128, size=(N, K)).astype(np.int8)
        codes = w_user
    elif source_format == "int":
        if bit == 8:
            w_user = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
            codes = w_user
        elif bit == 1:
            w_user = rng.integers(0, 2, size=(N, K)).astype(np.int8)   # codes; int1 decodes to {0,-1}
            codes = w_user
        else:
            maxq = 1 << (bit - 1)
            w_user = rng.integers(-maxq, maxq, size=(N, K), dtype=np.int8) if big else rng.integers(-maxq, maxq, size=(N, K)).astype(np.int8)
            codes = (w_user + maxq).astype(np.int8)
    elif source_format in ("nf", "fp"):
        w_user = rng.integers(0, 16, size=(N, K)).astype(np.int8)
        codes = w_user
    elif source_format in ("fp_e4m3", "fp_e5m2"):
        tdt = torch.float8_e4m3fn if source_format == "fp_e4m3" else torch.float8_e5m2
        wf = torch.from_numpy((rng.random((N, K), dtype=np.float32) * 2 - 1).astype(np.float32))
        w8 = wf.to(tdt)
        w_user = w8
        codes = w8.view(torch.int8).numpy()
    else:
        raise NotImplementedError(source_format)
    case["w_user"], case["codes"] = w_user, codes
    scale = zeros = bias = None
    if with_scaling:
        scale = (rng.random((N, K // g), dtype=np.float32) * scale_mul).astype(np.float16)
    if with_zeros:
        zval = float(1 << (bit - 1))
        if zeros_mode == "original":
            zeros = np.full((N, K // g), zval, dtype=np.float16)
            # make the zero points non-trivial but integer, like GPTQ checkpoints
            zeros = (zeros + rng.integers(-2, 2, size=zeros.shape)).astype(np.float16)
        elif zeros_mode == "rescale":
            zeros = (np.full((N, K // g), zval, dtype=np.float16) * scale).astype(np.float16)
        elif zeros_mode == "quantized":
            zint = np.clip(zval + rng.integers(-2, 2, size=(K // g, N)), 0, (1 << bit) - 1).astype(np.int8)
            zeros = oracle.general_compress(zint, bit)
    if with_bias:
        bdt = np.float16 if A_dtype == "float16" else np.int8
        bias = (rng.random((N,), dtype=np.float32)).astype(bdt) if A_dtype == "float16" else \
            rng.integers(-8, 8, size=(N,), dtype=np.int8)
    case.update(scale=scale, zeros=zeros, bias=bias)
    case["config"] = bitblas.MatmulConfig(
        M=M, N=N, K=K, A_dtype=A_dtype, W_dtype=W_dtype, out_dtype=out_dtype,
        accum_dtype=accum_dtype or ("int32" if A_dtype == "int8" else "float16"),
        layout="nt", with_bias=with_bias, group_size=group_size, with_scaling=with_scaling,
        with_zeros=with_zeros, zeros_mode=zeros_mode, fast_decoding=fast_decoding)
    case.update(A_dtype=A_dtype, out_dtype=out_dtype, zeros_mode=zeros_mode)
    return case


def oracle_output(case, strict_reference=None):
    """the oracle's result for a case.  strict_reference=None: the numerics of the last `hip_output` of this case (the library's
    default - strict_reference=False: IEEE e4m3, true unsigned uint8 - unless the test asked for the reference's quirks)"""
    if strict_reference is None:
        strict_reference = bool(case.get("_strict", False))
    return oracle.matmul_dequant(
        case["A"], case["codes"], source_format=case["source_format"], bit=case["bit"],
        scale=case["scale"], zeros=case["zeros"], zeros_mode=case["zeros_mode"],
        group_size=case["g"], bias=case["bias"], a_dtype=case["A_dtype"],
        out_dtype=case["out_dtype"], strict_reference=strict_reference)


def _to_dev(x, device):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.to(device)
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def hip_output(case, device="cuda", m_rows=None, matmul=None, strict_reference=None):
    """Run the product path: transform_weight (C packer) + Matmul.forward (HIP kernel).
    strict_reference=None: the operator as a caller of the reference constructs it (no extra argument: the library's default
    members - at M <= 2 the exact-product GEMV family); True: the reference's definition to the letter (per-element rounding
    members, e4m3 bit trick), for the tests that mean those members."""
    if matmul is not None:
        mm = matmul
        case["_strict"] = bool(getattr(mm, "strict_reference", False))
    elif strict_reference is None:
        mm = bitblas.Matmul(case["config"], enable_tuning=False)
        case["_strict"] = False
    else:
        mm = bitblas.Matmul(case["config"], enable_tuning=False, strict_reference=strict_reference)
        case["_strict"] = bool(strict_reference)
    w_user = case["w_user"]
    wt = w_user if isinstance(w_user, torch.Tensor) else torch.from_numpy(w_user)
    cfg = case["config"]
    if case["source_format"] == "int" and case["bit"] < 8 and (
            case["bit"] == 1 or cfg.with_scaling or cfg.with_zeros):
        # transform_weight clamps/offsets signed sources and (like the reference, general_matmul/
        # __init__.py:685-687) refuses int formats with scale/zeros; the reference's own op test
        # feeds `intweight + maxq` straight to weight_transform in that case
        # (test_general_matmul_ops_backend_tl.py:187-194).  int1 codes are fed pre-offset too.
        W = mm.weight_transform(torch.from_numpy(case["codes"])).to(device)
    else:
        W = mm.transform_weight(wt.to(device))
    A = _to_dev(case["A"], device)
    out = mm(A, W, scale=_to_dev(case["scale"], device), zeros=_to_dev(case["zeros"], device),
             bias=_to_dev(case["bias"], device))
    torch.cuda.synchronize()
    return out.cpu().numpy(), mm


def assert_fp_parity(got, want, rtol=1e-3, atol_frac=1e-3):
    """|got - want| <= rtol*|want| + atol, atol = atol_frac * rms(want): the north-star bound
    (1e-3 relative on fp16 outputs) with an absolute floor for outputs that cancel to ~0."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    assert np.isfinite(got).all()
    atol = atol_frac * max(float(np.sqrt(np.mean(want ** 2))), 1e-6)
    bad = np.abs(got - want) > rtol * np.abs(want) + atol
    assert not bad.any(), (
        f"{int(bad.sum())}/{bad.size} elements out of tolerance; max abs err "
        f"{np.abs(got - want).max():.4g}, rms(want) {np.sqrt(np.mean(want ** 2)):.4g}")


def contract(K, default_members=False, m=None, bf16=False, group_size=None, zeros_mode=None):
    """tolerances of include/wqaa.h's numerics contract (at `strict_reference`): 1e-3 relative + 1e-3 rms everywhere - the north
    star's bound - except where the default (exact-product) GEMV members at M <= 2 meet the TE definition's OWN per-element
    rounding, which they skip and which does not average out with few products per output: K < 4096, one group per row
    (per-channel scales) or `rescale` zero points get 2e-3 rms there (measured worst cases 1.43e-3 at K = 256, 1.6e-3 at
    K = 1024 per-channel, 1.4e-3 at K = 2112 with rescale; <= 9.3e-4 on the Llama-sized shapes, which therefore keep 1e-3:
    profiles/r05_parity_margins.txt, r06_parity_margins.txt).  bfloat16 results carry their own 2^-8 rounding.
    group_size / zeros_mode unknown (None): the looser bound only by K."""
    if bf16:
        return dict(rtol=8e-3, atol_frac=8e-3)
    exact_members = default_members and (m is None or m <= 2)
    few_products = K < 4096 or group_size in (-1, K) or zeros_mode == "rescale"
    return dict(rtol=1e-3, atol_frac=2e-3 if (exact_members and few_products) else 1e-3)


def case_contract(case, default_members=False, m=None):
    """`contract` of a `make_case` case: its K, group size and zeros mode"""
    zm = case["zeros_mode"] if case.get("zeros") is not None else None
    gs = -1 if case["g"] == case["K"] else case["g"]
    return contract(case["K"], default_members=default_members, m=case["M"] if m is None else m, group_size=gs, zeros_mode=zm)


_KNOB_VARS = {"gemv": "WQAA_GEMV_TUNE", "gemm": "WQAA_GEMM_TUNE", "two_pass": "WQAA_TWO_PASS"}


def set_knobs(monkeypatch, family, **kv):
    """tuning / test aids live in ONE variable per family as `key=value,key=value` (csrc/wqaa_common.h: knob): set or replace
    the given keys, keep the others; a value of None removes the key.  family: "gemv" | "gemm" | "two_pass"."""
    var = _KNOB_VARS[family]
    cur = {}
    for tok in filter(None, os.environ.get(var, "").split(",")):
        k, _, v = tok.partition("=")
        cur[k] = v
    for k, v in kv.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv(var, ",".join(f"{k}={v}" for k, v in cur.items()))
    else:
        monkeypatch.delenv(var, raising=False)


def knob_value(family, key):
    for tok in filter(None, os.environ.get(_KNOB_VARS[family], "").split(",")):
        k, _, v = tok.partition("=")
        if k == key:
            return v
    return None


def record_margin(tag, got, want):
    """achieved error of one parity case, appended to $WQAA_PARITY_MARGINS (tools/parity_margins.sh -> profiles/r04_parity_margins.txt):
    max |err| / |want| over the elements above 10 % of rms(want), and max |err| / rms(want) over all"""
    path = os.environ.get("WQAA_PARITY_MARGINS")
    if not path:
        return
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    rms = max(float(np.sqrt(np.mean(want ** 2))), 1e-30)
    err = np.abs(got - want)
    big = np.abs(want) > 0.1 * rms
    max_rel = float((err[big] / np.abs(want[big])).max()) if big.any() else 0.0
    with open(path, "a") as f:
        f.write(f"{tag}\tmax_rel={max_rel:.3e}\tmax_abs_over_rms={float(err.max()) / rms:.3e}\tn={got.size}\n")
