"""Members a plain `Matmul` call on a swept shape does not reach (tests/test_member_coverage_gpu.py covers those): the instantiations behind
the other entry points and behind the tuning variables, each launched against the oracle.  What is here was found by the kernel census of
round 6 (tools/kernel_census.py: the library's kernels against the kernels a rocprofv3 run of the whole GPU suite launched):

* the in-kernel activation quantiser (`WQAA_EPI_QUANTIZE_INPUT`, integration/BitNet/utils_quant.py:157-168, :205-216) under every
  sub-byte weight format and both checkpoint layouts - tests/test_bitnet_gpu.py runs BitNet's own int2;
* `wqaa_dequantize` (B_decode, tirscript/matmul_dequantize_impl.py:391-449) for every format x mode x layout x 16-bit type -
  tests/test_two_pass_gpu.py runs thirteen of them;
* the mid-M member's instantiations (csrc/wqaa_gemm_mid_kernel.h) over layout x mode x tile height;
* the exact-product GEMV members at two activation rows and one weight row per wave, which the selector only takes under
  `WQAA_GEMV_TUNE=exact=2` (an A/B aid) for 2- and 1-bit weights."""
import ctypes

import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from bitblas_amd import lib as wlib
from helpers import _to_dev, assert_fp_parity, hip_output, make_case, oracle_output, set_knobs
from test_group_gpu import build

pytestmark = pytest.mark.gpu
DEV = "cuda"

MODES = {
    "none": dict(group_size=-1),
    "s": dict(group_size=128, with_scaling=True),
    "zo": dict(group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original"),
    "zr": dict(group_size=128, with_scaling=True, with_zeros=True, zeros_mode="rescale"),
    "zq": dict(group_size=128, with_scaling=True, with_zeros=True, zeros_mode="quantized"),
}


# ---- the in-kernel activation quantiser ------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(512, 2048), (1024, 8192), (96, 4096), (96, 16384)])     # one lane chunk per row (the four-row chunk tile) / several / few rows / 1-bit rows of two chunks
@pytest.mark.parametrize("wd,fd", [("int4", None), ("int4", True), ("uint4", None), ("int2", None), ("int2", False), ("uint2", None),
                                   ("int1", None), ("int1", False)])
def test_in_kernel_activation_quantiser_every_weight_format(wd, fd, N, K, m):
    """x (float16) -> per-token int8 in the staging pass -> W_q x A_int8 -> `/ si / sw -> half (+ bias)`: one launch, bit for bit the
    oracle's restatement of BitLinearBitBLAS.forward with this operator's integer weights"""
    rng = np.random.default_rng(m + N + K + len(wd) + 7 * bool(fd))
    src, bit = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[wd]
    cfg = bitblas.MatmulConfig(M=[1, 16], N=N, K=K, A_dtype="int8", W_dtype=wd, out_dtype="float16", accum_dtype="int32", with_bias=True,
                               fast_decoding=fd)
    mm = bitblas.Matmul(cfg, enable_tuning=False)
    codes = rng.integers(0, 1 << bit, size=(N, K)).astype(np.int8)
    W = mm.weight_transform(torch.from_numpy(codes)).cuda()
    bias = rng.standard_normal(N).astype(np.float16)
    x = (rng.standard_normal((m, K)) * 2).astype(np.float16)
    x[0, :16] = 0
    sw = 3.25
    out = torch.empty((m, N), dtype=torch.float16, device=DEV)
    xd, bd = torch.from_numpy(x).cuda(), torch.from_numpy(bias).cuda()
    mm.lib.run_fused_quant(xd.data_ptr(), W.data_ptr(), bd.data_ptr(), out.data_ptr(), m, wlib.current_stream_handle(xd.device), sw)
    torch.cuda.synchronize()
    wq = oracle.dequantize_weight(codes, src, bit, a_dtype="int8")
    want = oracle.bitnet_forward(x, wq, np.float32(sw), bias)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), want.view(np.uint16))


# ---- B_decode on its own ----------------------------------------------------------------------------------------------------
def _dequant_cases():
    out = []
    for a in ("float16", "bfloat16"):
        for wd in ("uint4", "int4", "uint2", "int2", "uint1", "int1", "uint8", "int8", "nf4", "fp4_e2m1", "e4m3_float8", "e5m2_float8"):
            if a == "bfloat16" and wd == "e5m2_float8":
                continue
            subbyte = wd[0] in "ui" and wd not in ("uint8", "int8")
            modes = ["none", "s"] + (["zo", "zr", "zq"] if wd.startswith("uint") else [])
            if wd == "fp4_e2m1":
                modes = ["none"]
            for mode in modes:
                for fd in ([None, False] if (subbyte and a == "float16") else [None]):
                    for strict in ((True, False) if wd == "e4m3_float8" and a == "float16" else (True,)):
                        out.append(pytest.param(a, wd, mode, fd, strict, id=f"{a}-{wd}-{mode}-{'auto' if fd is None else 'plain'}{'' if strict else '-ieee'}"))
    return out


def _dequant_operands(a, wd, mode, seed):
    """codes + Scale / Zeros in the activation type (bfloat16 operators keep their metadata in bfloat16: tests/test_gemm_gpu.py _bf16_case)"""
    N, K = 272, 1024
    rng = np.random.default_rng(seed)
    src, bit = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[wd]
    tdt = torch.float16 if a == "float16" else torch.bfloat16
    if src in ("fp_e4m3", "fp_e5m2"):
        w8 = torch.from_numpy(rng.random((N, K), dtype=np.float32) * 2 - 1).to(torch.float8_e4m3fn if src == "fp_e4m3" else torch.float8_e5m2)
        codes = w8.view(torch.int8).numpy()
    elif src == "int" and bit == 8:
        codes = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    else:
        codes = rng.integers(0, 128 if bit == 8 else 1 << bit, size=(N, K)).astype(np.int8)
    kw = dict(MODES[mode])
    g = kw.get("group_size", -1)
    gg = K if g == -1 else g
    scale = zeros = None
    if kw.get("with_scaling"):
        scale = torch.from_numpy(rng.random((N, K // gg), dtype=np.float32) * 0.05).to(tdt)
    zm = kw.get("zeros_mode")
    if zm == "quantized":
        zint = np.clip((1 << (bit - 1)) + rng.integers(-2, 2, size=(K // gg, N)), 0, (1 << bit) - 1).astype(np.uint8).view(np.int8)
        zeros = oracle.general_compress(zint, bit)
    elif zm in ("original", "rescale"):
        zt = torch.from_numpy(((1 << (bit - 1)) + rng.integers(-2, 3, size=(N, K // gg))).astype(np.float32)).to(tdt)
        zeros = (zt.float() * scale.float()).to(tdt) if zm == "rescale" else zt
    return N, K, src, bit, gg, codes, scale, zeros, kw


@pytest.mark.parametrize("a,wd,mode,fd,strict", _dequant_cases())
def test_dequantize_every_member_bit_for_bit(a, wd, mode, fd, strict):
    N, K, src, bit, gg, codes, scale, zeros, kw = _dequant_operands(a, wd, mode, seed=len(a + wd + mode))
    cfg = bitblas.MatmulConfig(M=16, N=N, K=K, A_dtype=a, W_dtype=wd, accum_dtype="float32", out_dtype=a, fast_decoding=fd, **kw)
    mm = bitblas.Matmul(cfg, enable_tuning=False, strict_reference=strict)
    W = mm.weight_transform(torch.from_numpy(codes)).cuda() if mm.weight_transform is not None else torch.from_numpy(codes).cuda()
    tdt = bitblas.matmul.torch_dtype(a)
    out = torch.empty((N, K), dtype=tdt, device=DEV)
    lut = mm._ensure_lut(torch.device(DEV, torch.cuda.current_device()))
    sd = None if scale is None else scale.cuda()
    zd = None if zeros is None else (zeros.cuda() if isinstance(zeros, torch.Tensor) else torch.from_numpy(zeros).cuda())
    L = wlib.load_library()
    wlib.check(L.wqaa_dequantize(ctypes.byref(mm.lib.desc), W.data_ptr(), lut.data_ptr() if lut is not None else None,
                                 sd.data_ptr() if sd is not None else None, zd.data_ptr() if zd is not None else None,
                                 out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    lut_np = np.asarray(bitblas.Matmul.NF4_VALUES, dtype=np.float16) if src == "nf" else None
    want = oracle.dequantize_weight(codes, src, bit, scale=None if scale is None else scale.float().numpy(),
                                    zeros=zeros.float().numpy() if isinstance(zeros, torch.Tensor) else zeros,
                                    zeros_mode=kw.get("zeros_mode", "original"), group_size=gg, a_dtype=a, strict_reference=strict, lut=lut_np)
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    wbits = torch.from_numpy(np.asarray(want, dtype=np.float32)).to(tdt).view(torch.int16).numpy().view(np.uint16)
    bad = int((got != wbits).sum())
    assert bad == 0, f"{bad} of {got.size} elements differ"


# ---- the mid-M member: layout x mode x tile height ------------------------------------------------------------------------
@pytest.mark.parametrize("M", [32, 64, 128])
@pytest.mark.parametrize("fd", [None, False], ids=["lop3", "plain"])
@pytest.mark.parametrize("mode", list(MODES))
def test_mid_member_every_layout_and_mode(mode, fd, M, monkeypatch):
    set_knobs(monkeypatch, "gemm", mid=2)
    case = make_case(M, 1024, 4096, W_dtype="uint4", fast_decoding=fd, scale_mul=0.02, seed=M + len(mode), **MODES[mode])
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xmk"), mm.plans[M]["name"]
    assert_fp_parity(got, oracle_output(case))


# ---- exact-product GEMV, two activation rows, one weight row per wave (A/B aid) -------------------------------------------
@pytest.mark.parametrize("bits", [4, 2, 1])
@pytest.mark.parametrize("fd", [None, False], ids=["lop3", "plain"])
@pytest.mark.parametrize("mode", list(MODES))
def test_exact_members_at_two_rows_under_the_ab_switch(mode, fd, bits, monkeypatch):
    set_knobs(monkeypatch, "gemv", exact=2)
    M, N, K = 2, 384, 4096
    case = make_case(M, N, K, W_dtype=f"uint{bits}", fast_decoding=fd, scale_mul=0.05, seed=bits + len(mode), **MODES[mode])
    mm = bitblas.Matmul(case["config"], enable_tuning=False)
    assert "gemvx_b2r1" in mm.plans[M]["name"], mm.plans[M]["name"]
    W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()
    out = mm(torch.from_numpy(case["A"]).cuda(), W, scale=_to_dev(case["scale"], DEV), zeros=_to_dev(case["zeros"], DEV))
    torch.cuda.synchronize()
    want = oracle.matmul_dequant_exact(case["A"], case["codes"], source_format=case["source_format"], bit=bits, scale=case["scale"], zeros=case["zeros"],
                                       zeros_mode=case["zeros_mode"], group_size=case["g"], out_dtype="float16")
    assert_fp_parity(out.cpu().numpy(), want, rtol=1e-3, atol_frac=6e-4)


# ---- twins the selector takes on large shapes only, reached here through the A/B switches ---------------------------------
@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("wd,fd,N,K", [("int2", False, 2048, 8192), ("int2", None, 2048, 8192), ("int1", False, 2048, 16384), ("int1", None, 2048, 16384),
                                       ("int4", None, 2048, 4096)])
def test_lds_staged_twins_of_the_register_resident_int8_members(wd, fd, N, K, M, monkeypatch):
    """W_q x A_int8 rows of two lane chunks: the selector keeps the activation slice in registers up to 10 waves per CU of rows and
    stages it through LDS beyond (N > 5120: csrc/wqaa_gemv.hip choose); WQAA_GEMV_TUNE=areg=0 asks for the LDS-staged twin at a size
    the oracle takes in a second.  Integer sums: bit exact."""
    set_knobs(monkeypatch, "gemv", areg=0)
    case = make_case(M, N, K, W_dtype=wd, A_dtype="int8", out_dtype="int32", fast_decoding=fd, seed=M + K)
    got, mm = hip_output(case)
    assert "_areg" not in mm.plans[M]["name"] and "gemv_b" in mm.plans[M]["name"], mm.plans[M]["name"]
    assert np.array_equal(got, oracle_output(case))


@pytest.mark.parametrize("mode", list(MODES))
def test_exact_member_with_register_resident_activations_two_rows_per_wave(mode, monkeypatch):
    """4-bit LOP3 weights, K within one step: the selector keeps the activations in registers for N <= 2048 (one row per wave) and
    N >= 24576 (two rows per wave); WQAA_GEMV_TUNE=areg=1 asks for the latter at 8192 rows"""
    set_knobs(monkeypatch, "gemv", areg=1)
    M, N, K = 1, 8192, 4096
    case = make_case(M, N, K, W_dtype="uint4", scale_mul=0.05, seed=len(mode), **MODES[mode])
    mm = bitblas.Matmul(case["config"], enable_tuning=False)
    assert "gemvx_b1r2" in mm.plans[M]["name"] and mm.plans[M]["name"].endswith("_areg"), mm.plans[M]["name"]
    W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()
    out = mm(torch.from_numpy(case["A"]).cuda(), W, scale=_to_dev(case["scale"], DEV), zeros=_to_dev(case["zeros"], DEV))
    torch.cuda.synchronize()
    want = oracle.matmul_dequant_exact(case["A"], case["codes"], source_format=case["source_format"], bit=4, scale=case["scale"], zeros=case["zeros"],
                                       zeros_mode=case["zeros_mode"], group_size=case["g"], out_dtype="float16")
    assert_fp_parity(out.cpu().numpy(), want, rtol=1e-3, atol_frac=6e-4)
