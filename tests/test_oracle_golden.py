"""The oracle against vectors produced by the reference's own numpy helpers (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

import wqaa_oracle as oracle

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_compress_matches_reference(golden):
    n = 0
    for key in golden.files:
        if key.startswith("compress_"):
            tag = key[len("compress_"):]
            bits = int(tag[1])
            mine = oracle.general_compress(golden["codes_" + tag], bits)
            assert np.array_equal(mine, golden[key]), key
            assert np.array_equal(oracle.general_decompress(mine, bits), golden["codes_" + tag])
            n += 1
    assert n >= 9


def test_interleave_matches_reference(golden):
    n = 0
    for key in golden.files:
        if key.startswith("interleave_"):
            _, tgt, tag = key.split("_", 2)
            bits = int(tag[1])
            mine = oracle.interleave_weight(golden["compress_" + tag], bits, tgt, follow="numpy")
            assert np.array_equal(mine, golden[key]), key
            n += 1
    # 4b/f16, 4b/i8, 2b/i8 from quantization/utils.py; 2b/f16 and 1b/i8 from the reference test suite's own copy of
    # the helper (the library copy crashes under numpy 2), see oracle/gen_golden.py
    assert n >= 15


def test_signed_source_offset(golden):
    codes = oracle.weight_to_codes(golden["int4_signed_src"], "int", 4)
    assert np.array_equal(oracle.general_compress(codes, 4), golden["int4_signed_compress"])


def test_interleave_is_a_bit_permutation():
    rng = np.random.default_rng(3)
    for bits in (1, 2, 4):
        for tgt in ("float16", "int8"):
            x = rng.integers(-128, 128, size=(5, 64), dtype=np.int8)
            y = oracle.interleave_weight(x, bits, tgt)
            assert np.array_equal(oracle.deinterleave_weight(y, bits, tgt), x)
            assert np.unpackbits(x.view(np.uint8)).sum() == np.unpackbits(y.view(np.uint8)).sum()


def test_decoders_known_values():
    # int4 code u -> u - 8 ; int2 -> u - 2 ; int1 -> {0,-1}  (quantization.py:185-230)
    assert list(oracle.decode_codes(np.arange(16), "int", 4)) == [float(u - 8) for u in range(16)]
    assert list(oracle.decode_codes(np.arange(4), "int", 2)) == [-2.0, -1.0, 0.0, 1.0]
    assert list(oracle.decode_codes(np.arange(2), "int", 1)) == [0.0, -1.0]
    # fp4: sign + 3 exponent bits (quantization.py:141-156)
    fp4 = oracle.decode_codes(np.arange(16), "fp", 4)
    assert fp4[0] == 0.0 and fp4[8] == 0.0
    assert fp4[1] == 2.0 ** -6 and fp4[7] == 1.0 and fp4[15] == -1.0
    # e4m3 trick: exact on normals, zero -> 2^-7 (quantization.py:169-176)
    import torch
    vals = torch.tensor([1.0, -1.5, 0.015625, 448.0, 0.0]).to(torch.float8_e4m3fn)
    strict = oracle.decode_codes(vals.view(torch.int8).numpy(), "fp_e4m3", 8, True)
    ieee = oracle.decode_codes(vals.view(torch.int8).numpy(), "fp_e4m3", 8, False)
    assert list(ieee) == [1.0, -1.5, 0.015625, 448.0, 0.0]
    assert list(strict[:4]) == [1.0, -1.5, 0.015625, 448.0] and strict[4] == 2.0 ** -7
    # all e4m3 bytes: IEEE decode agrees with torch
    allb = np.arange(256, dtype=np.uint8)
    t = torch.from_numpy(allb.copy()).view(torch.float8_e4m3fn).float().numpy()
    mine = oracle.decode_e4m3_ieee(allb)
    ok = np.isnan(t) | (t == mine)
    assert ok.all()
    t5 = torch.from_numpy(allb.copy()).view(torch.float8_e5m2).float().numpy()
    m5 = oracle.decode_e5m2(allb)
    assert (np.isnan(t5) | (t5 == m5)).all()


def test_matmul_oracle_against_plain_float_math():
    rng = np.random.default_rng(0)
    M, N, K, g = 3, 8, 64, 32
    A = (rng.random((M, K)) - 0.5).astype(np.float16)
    codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
    scale = rng.random((N, K // g)).astype(np.float16)
    zeros = np.full((N, K // g), 8, dtype=np.float16)
    out = oracle.matmul_dequant(A, codes, source_format="uint", bit=4, scale=scale, zeros=zeros,
                                zeros_mode="original", group_size=g)
    w = ((codes.astype(np.float16) - np.repeat(zeros, g, 1)).astype(np.float16) * np.repeat(scale, g, 1)).astype(np.float16)
    ref = (A.astype(np.float64) @ w.astype(np.float64).T).astype(np.float16)
    assert np.array_equal(out, ref)
    # quantized zeros == original zeros when the zero points are the same integers
    zq = oracle.general_compress(np.full((K // g, N), 8, dtype=np.int8), 4)
    out_q = oracle.matmul_dequant(A, codes, source_format="uint", bit=4, scale=scale, zeros=zq,
                                  zeros_mode="quantized", group_size=g)
    assert np.array_equal(out_q, out)
    # bias is added after the cast
    bias = rng.random(N).astype(np.float16)
    out_b = oracle.matmul_dequant(A, codes, source_format="uint", bit=4, scale=scale, zeros=zeros,
                                  group_size=g, bias=bias)
    assert np.array_equal(out_b, (out + bias).astype(np.float16))


def test_gptq_unpack_helpers():
    rng = np.random.default_rng(1)
    z = rng.integers(0, 15, size=(4, 16)).astype(np.int64)   # stored zero points (value - 1)
    packed = np.zeros((4, 2), dtype=np.int64)
    for c in range(16):
        packed[:, c // 8] |= z[:, c] << (4 * (c % 8))
    q = packed.astype(np.uint32).view(np.int32)
    assert np.array_equal(oracle.unpack_qzeros(q, 4), (z + 1) & 15)
    assert np.array_equal(oracle.unpack_qzeros(q, 4, v2=True), z)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_bitnet_caller_ops_against_reference_run_vectors(tag):
    """oracle/gen_bitnet_golden.py RUNS the reference's BitLinearBitBLAS.weight_quant / activation_quant /
    post_quant_process (integration/BitNet/utils_quant.py:155-176) on seeded inputs; the oracle's restatement
    must reproduce every intermediate and the final half output bit for bit."""
    g = np.load(os.path.join(GOLDEN_DIR, "bitnet_golden.npz"))
    W, x = g[f"{tag}_W"], g[f"{tag}_x"]
    wq, sw = oracle.bitnet_weight_quant(W)
    assert np.array_equal(wq, g[f"{tag}_wq"])
    assert np.float32(sw) == g[f"{tag}_sw"]
    q, si = oracle.bitnet_activation_quant(x)
    assert np.array_equal(q, g[f"{tag}_q"])
    assert np.array_equal(si, g[f"{tag}_si"])
    bias = g[f"{tag}_bias"] if f"{tag}_bias" in g.files else None
    y = oracle.bitnet_forward(x, wq, sw, bias)
    assert np.array_equal(y.view(np.uint16), g[f"{tag}_y"].view(np.uint16))


def test_int4_activations_against_the_reference_tests_expectation():
    """oracle/gen_int4_golden.py RUNS testing/python/operators/test_general_matmul_ops_int4.py's
    matmul_int4_torch_forward (int4 x int4, int4 x int2): the packed operands exactly as that test hands them to
    the operator, and its expected `A.float() @ B.T.float()`.  Pins nibble / field order of both operands."""
    g = np.load(os.path.join(GOLDEN_DIR, "int4_golden.npz"))
    seen = set()
    for i in range(2):
        w_bits = {"int4": 4, "int2": 2}[str(g[f"c{i}_W_dtype"])]
        seen.add(w_bits)
        codes = oracle.general_decompress(g[f"c{i}_B"], w_bits)
        got = oracle.matmul_int4_act(g[f"c{i}_A"], codes, w_bits=w_bits)
        assert np.array_equal(got, g[f"c{i}_expected"])
    assert seen == {2, 4}


def test_oracle_properties_exact():
    """size-independent properties of the restated semantics that hold bit for bit: doubling every scale doubles the
    output (power-of-two factors commute with every fp16 / fp32 rounding short of overflow), integer activations are
    linear, all-zero-point weights give the bias."""
    rng = np.random.default_rng(5)
    N, K, g = 48, 256, 64
    A = (rng.random((5, K), dtype=np.float32) - 0.5).astype(np.float16)
    codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
    scale = (rng.random((N, K // g), dtype=np.float32) * 0.05 + 0.01).astype(np.float16)
    zeros = rng.integers(6, 10, size=(N, K // g)).astype(np.float16)
    kw = dict(source_format="uint", bit=4, zeros=zeros, zeros_mode="original", group_size=g, out_dtype="float32")
    y1 = oracle.matmul_dequant(A, codes, scale=scale, **kw)
    y2 = oracle.matmul_dequant(A, codes, scale=(scale * np.float16(2)).astype(np.float16), **kw)
    assert np.array_equal(y2, y1 * np.float32(2))
    # integer path: exact linearity in the activations
    A1 = rng.integers(-60, 60, size=(3, K), dtype=np.int8)
    A2 = rng.integers(-60, 60, size=(3, K), dtype=np.int8)
    c2 = rng.integers(0, 4, size=(N, K)).astype(np.int8)
    f = lambda a: oracle.matmul_dequant(a, c2, source_format="int", bit=2, a_dtype="int8", out_dtype="int32")
    assert np.array_equal(f((A1 + A2).astype(np.int8)), f(A1) + f(A2))
    # weights sitting on their zero point contribute nothing: the output is the bias (added after the cast)
    bias = rng.random(N, dtype=np.float32).astype(np.float16)
    flat = np.full((N, K), 8, dtype=np.int8)
    y0 = oracle.matmul_dequant(A, flat, source_format="uint", bit=4, scale=scale, zeros=np.full((N, K // g), 8, np.float16),
                               zeros_mode="original", group_size=g, bias=bias, out_dtype="float16")
    assert np.array_equal(y0, np.broadcast_to(bias, y0.shape))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_layer_elementwise_ops_against_reference_run_vectors(tag):
    """The decoder layer's RMSNorm, gated activation and residual add (the ops that ride in the GEMV launches): the oracle's
    restatements against vectors produced by RUNNING the reference's `BitnetRMSNorm` class and its layer expressions
    (oracle/gen_layer_ops_golden.py -> tests/golden/layer_ops_golden.npz).  Norm and add: bit for bit.  silu * up: numpy's exp and
    torch's round the last fp32 bit differently on a few inputs in ten thousand - those land one float16 ulp away, nothing more."""
    g = np.load(os.path.join(GOLDEN_DIR, "layer_ops_golden.npz"))
    y = oracle.rms_norm_f16(g[f"norm_{tag}_x"], g[f"norm_{tag}_w"], float(g[f"norm_{tag}_eps"]))
    assert np.array_equal(y.view(np.uint16), g[f"norm_{tag}_y"].view(np.uint16))
    r = oracle.add_residual_f16(g[f"add_{tag}_hidden"], g[f"add_{tag}_residual"])
    assert np.array_equal(r.view(np.uint16), g[f"add_{tag}_y"].view(np.uint16))
    a, want = oracle.silu_mul_f16(g[f"act_{tag}_gate"], g[f"act_{tag}_up"]), g[f"act_{tag}_y"]
    off = a.view(np.uint16) != want.view(np.uint16)
    assert off.sum() <= max(1, a.size // 2000)
    assert np.all(np.abs(a[off].astype(np.float64) - want[off].astype(np.float64)) <= np.abs(want[off].astype(np.float64)) * 2.0 ** -9)
