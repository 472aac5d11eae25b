"""Decode known-answer tests on the GPU: pack -> (interleave) -> device decode -> equals the source.

HIP twin of the reference's native gtest suites (testing/cpp/lop3_type_conversion/
lowprecision_to_float16.cu:51-99, lowprecision_to_int8.cu:139-240): same structure, but values
follow the Python/TE convention (int formats subtract 2^(bits-1)).
"""
import ctypes

import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
from bitblas_amd import lib as wlib

pytestmark = pytest.mark.gpu


def device_decode(packed_np, w_format, bits, layout, a_code, strict=1, lut=None):
    L = wlib.load_library()
    words = torch.from_numpy(packed_np.view(np.int32).copy()).cuda().contiguous()
    n = words.numel()
    epw = 32 // bits
    out = torch.empty(n * epw, dtype=torch.float16 if a_code == wlib.F16 else torch.int8, device="cuda")
    lut_t = None if lut is None else torch.tensor(lut, dtype=torch.float16, device="cuda")
    st = L.wqaa_debug_decode(words.data_ptr(), n, w_format, bits, layout, a_code, strict,
                             None if lut_t is None else lut_t.data_ptr(), out.data_ptr(), None)
    wlib.check(st)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("bits", [4, 2, 1])
@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("layout", [wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3])
@pytest.mark.parametrize("a_code", [wlib.F16, wlib.I8])
def test_integer_decode(bits, signed, layout, a_code):
    rng = np.random.default_rng(bits * 10 + signed)
    codes = rng.integers(0, 1 << bits, size=(16, 512), dtype=np.int8)
    packed = oracle.general_compress(codes, bits)
    if layout == wlib.LAYOUT_LOP3:
        packed = oracle.interleave_weight(packed, bits, "float16" if a_code == wlib.F16 else "int8")
    got = device_decode(packed, wlib.W_INT if signed else wlib.W_UINT, bits, layout, a_code)
    want = codes.astype(np.int32) - ((1 << (bits - 1)) if signed else 0)
    if signed and bits == 1:
        want = -codes.astype(np.int32)   # int1 is sign-extended: {0, -1} (quantization.py:220-230)
    assert np.array_equal(got.astype(np.int32).reshape(codes.shape), want)


@pytest.mark.parametrize("signed", [False, True])
def test_int8_weight_to_f16(signed):
    rng = np.random.default_rng(5)
    w = rng.integers(-128 if signed else 0, 128 if signed else 256, size=(4, 256)).astype(np.int16)
    packed = w.astype(np.uint8 if not signed else np.int8).view(np.int8)
    got = device_decode(packed, wlib.W_INT if signed else wlib.W_UINT, 8, wlib.LAYOUT_PLAIN, wlib.F16)
    assert np.array_equal(got.astype(np.int32).reshape(w.shape), w.astype(np.int32))


def test_nf4_and_fp4_lut_decode():
    codes = np.tile(np.arange(16, dtype=np.int8), 64).reshape(8, 128)
    packed = oracle.general_compress(codes, 4)
    lut = oracle.NF4_LUT.astype(np.float16)
    got = device_decode(packed, wlib.W_NF, 4, wlib.LAYOUT_PLAIN, wlib.F16, lut=lut.tolist())
    assert np.array_equal(got.reshape(codes.shape), lut[codes])
    got4 = device_decode(packed, wlib.W_FP4, 4, wlib.LAYOUT_PLAIN, wlib.F16)
    assert np.array_equal(got4.reshape(codes.shape).astype(np.float64), oracle.decode_fp4(codes))


@pytest.mark.parametrize("strict", [1, 0])
def test_e4m3_decode_all_bytes(strict):
    allb = np.arange(256, dtype=np.uint8).view(np.int8)
    got = device_decode(allb, wlib.W_E4M3, 8, wlib.LAYOUT_PLAIN, wlib.F16, strict=strict).astype(np.float64)
    want = oracle.decode_e4m3_strict(allb) if strict else oracle.decode_e4m3_ieee(allb)
    ok = (got == want) | np.isnan(want)      # 0x7f/0xff are NaN in OCP e4m3: outside the contract
    assert ok.all(), np.nonzero(~ok)


def test_e5m2_decode_all_bytes():
    allb = np.arange(256, dtype=np.uint8).view(np.int8)
    got = device_decode(allb, wlib.W_E5M2, 8, wlib.LAYOUT_PLAIN, wlib.F16).astype(np.float64)
    want = oracle.decode_e5m2(allb)
    assert ((got == want) | (np.isnan(want) & np.isnan(got))).all()
