"""The C ABI: the library loads without a GPU and exports every symbol include/wqaa.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from bitblas_amd import lib as wlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "wqaa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:void|int|char\s*\*|const char\s*\*)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_and_binding_agree():
    names = declared_functions()
    assert "init" in names and "wqaa_matmul" in names
    assert sorted(wlib.EXPORTED_SYMBOLS) == names


def test_library_exports_every_declared_symbol():
    cdll = ctypes.CDLL(wlib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(cdll, name), f"{name} missing from libwqaa_hip.so"


def test_struct_sizes_match_header():
    assert ctypes.sizeof(wlib.MatmulDesc) == 16 * 4
    assert ctypes.sizeof(wlib.Plan) == 11 * 4 + 96


def test_init_is_idempotent_and_error_channel_works():
    L = wlib.load_library()
    L.init()
    L.init()
    assert L.wqaa_abi_version() == 1
    d = wlib.make_desc(N=64, K=64, a_dtype=wlib.F16, w_format=wlib.W_UINT, w_bits=4, out_dtype=wlib.F16)
    d.struct_size = 4  # wrong ABI size must be refused
    assert L.wqaa_matmul(ctypes.byref(d), 1, 1, None, None, None, None, 1, 1, None) == wlib.ERR_BAD_DESC
    assert L.wqaa_last_error() == wlib.ERR_BAD_DESC
    assert b"ABI" in L.wqaa_last_error_string()


def test_m_zero_returns_ok_without_touching_anything():
    L = wlib.load_library()
    d = wlib.make_desc(N=64, K=64, a_dtype=wlib.F16, w_format=wlib.W_UINT, w_bits=4, out_dtype=wlib.F16)
    assert L.wqaa_matmul(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == wlib.OK


def test_selector_answers_without_gpu():
    p1 = wlib.select(wlib.make_desc(N=4096, K=4096, a_dtype=wlib.F16, w_format=wlib.W_INT, w_bits=4,
                                    out_dtype=wlib.F16, group_size=128, with_scaling=True,
                                    w_layout=wlib.LAYOUT_LOP3), 1)
    assert p1["kernel_family"] == 1 and p1["grid"] >= 256 and "gemv" in p1["name"]
    with pytest.raises(wlib.WqaaError) as e:
        wlib.select(wlib.make_desc(N=64, K=40, a_dtype=wlib.F16, w_format=wlib.W_INT, w_bits=4,
                                   out_dtype=wlib.F16), 1)
    assert e.value.code == wlib.ERR_UNSUPPORTED


def test_unsupported_is_loud_not_silent():
    with pytest.raises(wlib.WqaaError):
        wlib.select(wlib.make_desc(N=64, K=64, a_dtype=wlib.F32, w_format=wlib.W_INT, w_bits=4,
                                   out_dtype=wlib.F16), 1)


def test_c_packer_against_oracle_and_golden(golden):
    import wqaa_oracle as oracle
    rng = np.random.default_rng(0)
    for bits in (1, 2, 4):
        for code, tgt in ((wlib.F16, "float16"), (wlib.I8, "int8")):
            codes = rng.integers(0, 1 << bits, size=(9, 128), dtype=np.int8)
            plain = wlib.pack_weight(codes, bits, wlib.LAYOUT_PLAIN, code)
            assert np.array_equal(plain, oracle.general_compress(codes, bits))
            lop3 = wlib.pack_weight(codes, bits, wlib.LAYOUT_LOP3, code)
            assert np.array_equal(lop3, oracle.interleave_weight(plain, bits, tgt))
            assert np.array_equal(wlib.unpack_weight(lop3, 128, bits, wlib.LAYOUT_LOP3, code), codes)
    # int4 activations: 2-bit weights interleaved for a 4-bit target (lop3_permutate_impl.py:27-35, 131-134)
    codes = rng.integers(0, 4, size=(9, 128), dtype=np.int8)
    plain = wlib.pack_weight(codes, 2, wlib.LAYOUT_PLAIN, wlib.I4)
    lop3 = wlib.pack_weight(codes, 2, wlib.LAYOUT_LOP3, wlib.I4)
    assert np.array_equal(lop3, oracle.interleave_weight(plain, 2, "int4"))
    assert np.array_equal(wlib.unpack_weight(lop3, 128, 2, wlib.LAYOUT_LOP3, wlib.I4), codes)
    for key in golden.files:
        if key.startswith("interleave_"):
            _, tgt, tag = key.split("_", 2)
            bits = int(tag[1])
            code = wlib.F16 if tgt == "float16" else wlib.I8
            assert np.array_equal(wlib.pack_weight(golden["codes_" + tag], bits, wlib.LAYOUT_LOP3, code), golden[key]), key
        if key.startswith("compress_"):
            tag = key[len("compress_"):]
            bits = int(tag[1])
            assert np.array_equal(wlib.pack_weight(golden["codes_" + tag], bits, wlib.LAYOUT_PLAIN, wlib.F16), golden[key])
