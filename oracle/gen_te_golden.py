"""Golden vectors produced by EXECUTING the reference's own TE definitions and TIR decoders.

The operator the reference JIT-compiles is *defined* by two Python functions that build a TVM tensor-expression
graph: `matmul_nt_dequantize_b` (bitblas/ops/general_matmul/tirscript/matmul_dequantize_impl.py:339-499) and
`matmul_nt` (tirscript/matmul_impl.py:49-84), whose per-element decode calls the `_tir_*` helpers of
bitblas/quantization/quantization.py:141-230.  TVM is not installable here (its submodule is empty), but these
functions only use a small, purely functional slice of it: `te.placeholder / te.compute / te.reduce_axis / te.sum`
and `tir.const / Cast / reinterpret / Select / Min` with operator overloading on expressions.  This script provides
that slice as a numpy-backed interpreter ("a tensor expression evaluates to an array"), puts it in `sys.modules`
under the names the reference imports (`tvm`, `tvm.te`, `tvm.tir`, `bitblas`, `bitblas.quantization`, ...), and
then RUNS the reference's source files from where they lie under /root/reference - unmodified, via
`exec(compile(open(path).read(), path, "exec"))`.  What gets recorded per case:

    inputs   A, B (packed storage bytes, general_compress order), LUT, Scale, Zeros | QZeros, Bias  (seeded)
    B_decode the dequantised weight matrix exactly as the TE graph materialises it (in A_dtype)
    out      the graph's last stage (C -> cast to out_dtype -> + Bias)

Semantics the interpreter takes from TVM (tvm/src/tir/op/op.cc `BinaryOpMatchTypes`, stated here because they are
the only non-obvious part): mixed signed/unsigned operands of equal width compute in the unsigned type, unequal
widths in the wider type, integer with float in the float type; integer arithmetic wraps; `>>` on signed integers
is arithmetic; a Python int operand becomes a constant of the other operand's dtype; `astype` is a C-style cast
(float16 results are rounded to nearest even).  The reduction `te.sum` is evaluated in float64 / int64 and cast to
the accumulator dtype: the TE graph leaves the summation order open, so cases use float32 / int32 accumulators.
fp8 dtypes (`e4m3_float8`, `e5m2_float8`) are decoded with torch's float8 types - the reference's tests make the
same identification (testing/python/operators/test_general_matmul_fp8.py:20-33).

Output: tests/golden/te_golden.npz + te_golden.json (committed).  Runs only where /root/reference exists.
Test infrastructure - never imported by the product.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_NPZ = os.path.join(HERE, "..", "tests", "golden", "te_golden.npz")
OUT_JSON = os.path.join(HERE, "..", "tests", "golden", "te_golden.json")

NP = {"int8": np.int8, "uint8": np.uint8, "int16": np.int16, "uint16": np.uint16, "int32": np.int32,
      "uint32": np.uint32, "int64": np.int64, "uint64": np.uint64, "float16": np.float16, "float32": np.float32,
      "float64": np.float64, "bool": np.bool_}
FP8 = ("e4m3_float8", "e5m2_float8")


def _bits(dt):
    return int("".join(c for c in dt if c.isdigit())) if dt not in FP8 else 8


def _kind(dt):
    if dt in FP8 or dt.startswith("float"):
        return "f"
    if dt.startswith("uint"):
        return "u"
    if dt.startswith("int"):
        return "i"
    return "b"


def _narrow(dt):
    return _kind(dt) in "iu" and _bits(dt) < 32


def _fp8_to_f32(raw_u8, dt):
    import torch
    tdt = torch.float8_e4m3fn if dt == "e4m3_float8" else torch.float8_e5m2
    t = torch.from_numpy(np.ascontiguousarray(raw_u8).view(np.int8)).view(tdt)
    return t.to(torch.float32).numpy()


class Interp:
    """state of one evaluation: the arrays fed to placeholders (by name), every computed stage (by name), and the
    integer model: c_promotion=True evaluates sub-int operands as C does (the reference emits CUDA / HIP C through
    TVM's CodeGenC, where `(signed char)a << 10` is computed in int and narrows only at casts, reinterprets and
    stores); c_promotion=False evaluates every operation in its nominal TIR dtype (what an LLVM-typed lowering does)"""
    feeds: dict = {}
    stages: dict = {}
    const_overflow: bool = False
    c_promotion: bool = True


class E:
    """an evaluated expression: numpy array + TVM dtype name (fp8 values are carried as their raw bytes)"""
    __array_priority__ = 1000

    def __init__(self, v, dtype, _raw=False):
        self.dtype = dtype
        if _raw:
            self.v = v
            return
        a = np.asarray(v, dtype=np.uint8 if dtype in FP8 else NP[dtype])     # the value in its nominal type
        # C integer promotion (Interp.c_promotion): operands narrower than int are computed as int
        self.v = a.astype(np.int32) if (Interp.c_promotion and _narrow(dtype)) else a

    def nominal(self):
        """the value as stored in a variable of the nominal dtype (narrow integers wrap)"""
        if Interp.c_promotion and _narrow(self.dtype):
            return self.v.astype(NP[self.dtype])
        return self.v

    # ---- casts ----
    def astype(self, dtype):
        if dtype == self.dtype:
            return self                                   # TVM elides a cast to the same dtype
        src = _fp8_to_f32(self.nominal(), self.dtype) if self.dtype in FP8 else self.v
        if dtype in FP8:
            raise NotImplementedError("cast to fp8 is not used by the reference's definitions")
        with np.errstate(all="ignore"):
            return E(src.astype(NP[dtype]), dtype)

    # ---- binary ops ----
    @staticmethod
    def _lift(x, like):
        if isinstance(x, E):
            return x
        if isinstance(x, (bool, np.bool_)):
            return E(x, "bool")
        if isinstance(x, (int, np.integer)):
            return E(x, like.dtype if _kind(like.dtype) in "iu" else "int32") if _kind(like.dtype) != "f" else E(x, like.dtype)
        if isinstance(x, float):
            return E(x, like.dtype if _kind(like.dtype) == "f" else "float32")
        raise TypeError(type(x))

    @staticmethod
    def _match(a, b):
        if a.dtype == b.dtype:
            return a, b, a.dtype
        ka, kb = _kind(a.dtype), _kind(b.dtype)
        if ka == "f" and kb != "f":
            return a, b.astype(a.dtype), a.dtype
        if kb == "f" and ka != "f":
            return a.astype(b.dtype), b, b.dtype
        if ka == "f" and kb == "f":
            t = a.dtype if _bits(a.dtype) >= _bits(b.dtype) else b.dtype
            return a.astype(t), b.astype(t), t
        ba, bb = _bits(a.dtype), _bits(b.dtype)
        if ka == kb:
            t = a.dtype if ba >= bb else b.dtype
        elif ba < bb:
            t = b.dtype
        elif ba > bb:
            t = a.dtype
        else:
            t = a.dtype if ka == "u" else b.dtype      # equal width, mixed sign: the unsigned type
        return a.astype(t), b.astype(t), t

    def _bin(self, other, fn, result=None, reflected=False):
        o = E._lift(other, self)
        a, b = (o, self) if reflected else (self, o)
        a, b, t = E._match(a, b)
        with np.errstate(all="ignore"):
            r = fn(a.v, b.v)
        rt = result or t
        if Interp.c_promotion and _narrow(rt):
            return E(r.astype(np.int32), rt, _raw=True)   # no narrowing until a cast, a reinterpret or a store
        return E(r.astype(NP[rt]) if rt not in FP8 else r, rt)

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, reflected=True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, reflected=True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, reflected=True)
    def __and__(self, o): return self._bin(o, np.bitwise_and)
    def __rand__(self, o): return self._bin(o, np.bitwise_and, reflected=True)
    def __or__(self, o): return self._bin(o, np.bitwise_or)
    def __ror__(self, o): return self._bin(o, np.bitwise_or, reflected=True)
    def __xor__(self, o): return self._bin(o, np.bitwise_xor)
    def __rxor__(self, o): return self._bin(o, np.bitwise_xor, reflected=True)
    def __lshift__(self, o): return self._bin(o, np.left_shift)
    def __rshift__(self, o): return self._bin(o, np.right_shift)      # arithmetic for signed dtypes (numpy == C)
    def __eq__(self, o): return self._bin(o, np.equal, result="bool")        # noqa: E704
    def __ne__(self, o): return self._bin(o, np.not_equal, result="bool")
    def __gt__(self, o): return self._bin(o, np.greater, result="bool")
    def __ge__(self, o): return self._bin(o, np.greater_equal, result="bool")
    def __lt__(self, o): return self._bin(o, np.less, result="bool")
    def __le__(self, o): return self._bin(o, np.less_equal, result="bool")
    __hash__ = None

    def __floordiv__(self, o):
        return self._bin(o, np.floor_divide)

    def __mod__(self, o):
        return self._bin(o, np.mod)


class Tensor:
    def __init__(self, name, shape, dtype, data=None):
        self.name, self.shape, self.dtype, self.data = name, tuple(shape), dtype, data

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        assert self.data is not None, f"placeholder {self.name} was read but never fed"
        arrs = [i.v.astype(np.int64) if isinstance(i, E) else np.asarray(i, dtype=np.int64) for i in idx]
        arrs = np.broadcast_arrays(*arrs)
        return E(self.data[tuple(arrs)], self.dtype)


def _placeholder(shape, name=None, dtype="float32"):
    data = Interp.feeds.get(name)
    if data is not None:
        want = np.uint8 if dtype in FP8 else NP[dtype]
        data = np.asarray(data)
        assert data.shape == tuple(shape), (name, data.shape, shape)
        data = data.view(want) if data.dtype.itemsize == np.dtype(want).itemsize else data.astype(want)
    return Tensor(name, shape, dtype, data)


class _ReduceAxis(E):
    pass


def _reduce_axis(dom, name=None):
    lo, hi = dom
    return _ReduceAxis(np.arange(lo, hi, dtype=np.int32), "int32")


def _sum(expr, axis=None):
    """reduction over the trailing (reduce) axis, order-free: int64 / float64 accumulation, cast to the expr dtype"""
    wide = np.int64 if _kind(expr.dtype) in "iu" else np.float64
    with np.errstate(all="ignore"):
        r = expr.v.astype(wide).sum(axis=-1, keepdims=True)
        return E(r.astype(NP[expr.dtype]), expr.dtype)


def _compute(shape, fcompute, name=None):
    nd = len(shape)
    idx = []
    for ax, n in enumerate(shape):
        sh = [1] * (nd + 1)                     # one trailing axis is reserved for a reduce axis
        sh[ax] = n
        idx.append(E(np.arange(n, dtype=np.int32).reshape(sh), "int32"))
    r = fcompute(*idx)
    v = r.nominal()
    if v.ndim == nd + 1:
        assert v.shape[-1] == 1, "a reduce axis survived the stage"
        v = v[..., 0]
    v = np.broadcast_to(v, tuple(shape)).copy()
    t = Tensor(name, shape, r.dtype, v)
    Interp.stages[name] = t
    return t


class _PrimFunc:
    def __init__(self, args):
        self.args = list(args)
        self.attrs = {}

    def with_attr(self, key, value):
        self.attrs[key] = value
        return self


def install_tvm_stand_in():
    tvm = types.ModuleType("tvm")
    te = types.ModuleType("tvm.te")
    tir = types.ModuleType("tvm.tir")
    te.placeholder, te.compute, te.reduce_axis, te.sum = _placeholder, _compute, _reduce_axis, _sum
    te.create_prim_func = lambda args: _PrimFunc(args)
    te.var = lambda name, dtype="int32": (_ for _ in ()).throw(NotImplementedError("dynamic M is a shape, not a value"))
    tir.PrimExpr = E

    def const(value, dtype="int32"):
        # TVM's IntImm range-checks its value in recent versions; a constant that does not fit (the 8-bit mask 255
        # as int8, quantization.py:202/214) is wrapped here like a C cast and the case is flagged
        if _kind(dtype) in "iu" and isinstance(value, (int, np.integer)):
            info = np.iinfo(NP[dtype])
            if not (info.min <= int(value) <= info.max):
                Interp.const_overflow = True
                value = np.array(int(value) & ((1 << info.bits) - 1), dtype=np.uint64).astype(NP[dtype])
        return E(value, dtype)

    tir.const = const
    tir.Cast = lambda dtype, value: value.astype(dtype)
    tir.Min = lambda a, b: a._bin(b, np.minimum)

    def reinterpret(dtype, value):
        assert _bits(dtype) == _bits(value.dtype), (dtype, value.dtype)
        raw = np.ascontiguousarray(value.nominal())
        return E(raw.view(np.uint8 if dtype in FP8 else NP[dtype]), dtype)

    def select(cond, a, b):
        a = E._lift(a, b if isinstance(b, E) else cond)
        b = E._lift(b, a)
        a, b, t = E._match(a, b)
        return E(np.where(cond.v, a.v, b.v), t, _raw=True)

    tir.reinterpret, tir.Select = reinterpret, select
    tir.IndexMap = object
    tvm.te, tvm.tir, tvm.DataType = te, tir, str
    tvm.IRModule = types.SimpleNamespace(from_expr=lambda f: f)

    bitblas = types.ModuleType("bitblas")
    bitblas.__path__ = []
    bitblas.tvm = tvm
    base = types.ModuleType("bitblas.base")
    base.__path__ = []
    gpu = types.ModuleType("bitblas.gpu")
    gpu.__path__ = []
    ma = types.ModuleType("bitblas.gpu.matmul_analysis")
    ma.get_propagate_map = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError)
    mods = {"tvm": tvm, "tvm.te": te, "tvm.tir": tir, "bitblas": bitblas, "bitblas.base": base, "bitblas.gpu": gpu,
            "bitblas.gpu.matmul_analysis": ma}
    sys.modules.update(mods)

    def run_reference_file(modname, relpath):
        path = os.path.join(REF, relpath)
        mod = types.ModuleType(modname)
        mod.__file__ = path
        sys.modules[modname] = mod
        exec(compile(open(path).read(), path, "exec"), mod.__dict__)
        return mod

    run_reference_file("bitblas.base.operator_common", "bitblas/base/operator_common.py")
    quant = run_reference_file("bitblas.quantization", "bitblas/quantization/quantization.py")
    deq = run_reference_file("ref_matmul_dequantize_impl", "bitblas/ops/general_matmul/tirscript/matmul_dequantize_impl.py")
    dense = run_reference_file("ref_matmul_impl", "bitblas/ops/general_matmul/tirscript/matmul_impl.py")
    return quant, deq, dense


def load_reference_utils():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_quant_utils", os.path.join(REF, "bitblas", "quantization", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


NF4 = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
       -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
       0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0]


def dequant_cases():
    """(tag, kwargs of matmul_nt_dequantize_b).  Shapes are small; the code matrix of every case cycles through ALL
    field values (all 256 bytes for 8-bit formats) before turning random, so each decoder is pinned exhaustively."""
    c = []
    f16 = dict(in_dtype="float16", out_dtype="float16", accum_dtype="float32")
    i8 = dict(in_dtype="int8", out_dtype="int32", accum_dtype="int32")
    for bit in (4, 2, 1):
        for fmt in ("uint", "int"):
            c.append((f"f16_{fmt}{bit}", dict(f16, bit=bit, source_format=fmt)))
            c.append((f"f16_{fmt}{bit}_scale_g32", dict(f16, bit=bit, source_format=fmt, with_scaling=True, group_size=32)))
            c.append((f"i8_{fmt}{bit}", dict(i8, bit=bit, source_format=fmt)))
    c.append(("i8_int2_bias", dict(i8, bit=2, source_format="int", with_bias=True)))
    for fmt in ("uint", "int"):
        c.append((f"f16_{fmt}8_scale", dict(f16, bit=8, source_format=fmt, with_scaling=True, group_size=64)))
        c.append((f"i8_{fmt}8", dict(i8, bit=8, source_format=fmt)))
    for zm in ("original", "rescale", "quantized"):
        for bit in (4, 2, 8):
            c.append((f"f16_uint{bit}_zeros_{zm}", dict(f16, bit=bit, source_format="uint", with_scaling=True, with_zeros=True,
                                                        group_size=32, zeros_mode=zm)))
    c.append(("f16_uint4_zeros_original_bias_f16acc_out", dict(in_dtype="float16", out_dtype="float16", accum_dtype="float32", bit=4,
                                                               source_format="uint", with_scaling=True, with_zeros=True,
                                                               group_size=64, zeros_mode="original", with_bias=True)))
    c.append(("f16_fp4", dict(f16, bit=4, source_format="fp")))
    c.append(("f16_fp4_scale", dict(f16, bit=4, source_format="fp", with_scaling=True, group_size=32)))
    c.append(("f16_e4m3", dict(f16, bit=8, source_format="fp_e4m3")))
    c.append(("f16_e4m3_scale", dict(f16, bit=8, source_format="fp_e4m3", with_scaling=True, group_size=32)))
    c.append(("f16_nf4", dict(f16, bit=4, source_format="nf")))
    c.append(("f16_nf4_scale", dict(f16, bit=4, source_format="nf", with_scaling=True, group_size=64)))
    c.append(("f16_uint4_out_f32", dict(in_dtype="float16", out_dtype="float32", accum_dtype="float32", bit=4, source_format="uint",
                                        with_scaling=True, group_size=-1)))
    return c


def build_inputs(tag, kw, ref_utils, rng, M=3, N=32, K=256):
    bit, fmt = kw["bit"], kw["source_format"]
    in_dt = kw["in_dtype"]
    g = K if kw.get("group_size", -1) == -1 else kw["group_size"]
    # codes: unsigned storage fields; first rows cycle through every value
    nvals = 1 << bit
    codes = rng.integers(0, nvals, size=(N, K)).astype(np.int64)
    cyc = np.arange(N * K, dtype=np.int64).reshape(N, K) % nvals
    codes[: max(1, (2 * nvals + K - 1) // K)] = cyc[: max(1, (2 * nvals + K - 1) // K)]
    if bit == 8:
        B = codes.astype(np.uint8).view(np.int8)
    else:
        B = ref_utils.general_compress(codes.astype(np.int8), source_bits=bit, storage_dtype=np.int8)
    feeds = {"B": B}
    if in_dt == "float16":
        A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
    else:
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    feeds["A"] = A
    if fmt == "nf":
        feeds["LUT"] = np.array(NF4, dtype=np.float16)
    if kw.get("with_scaling"):
        feeds["Scale"] = (rng.random((N, K // g), dtype=np.float32) * 0.1 + 0.01).astype(np.float16)
    if kw.get("with_zeros"):
        zm = kw["zeros_mode"]
        zint = np.clip((1 << (bit - 1)) + rng.integers(-3, 4, size=(K // g, N)), 0, nvals - 1)
        if zm == "quantized":
            feeds["QZeros"] = ref_utils.general_compress(zint.astype(np.int8), source_bits=bit, storage_dtype=np.int8) if bit < 8 \
                else zint.astype(np.uint8).view(np.int8)
        elif zm == "original":
            # integer zero points as GPTQ produces them, and a few non-integers (the TE subtraction then rounds)
            z = zint.T.astype(np.float16)
            z[::5, 0] += np.float16(0.37)
            feeds["Zeros"] = z
        else:
            feeds["Zeros"] = (zint.T.astype(np.float16) * feeds["Scale"]).astype(np.float16)
    if kw.get("with_bias"):
        feeds["Bias"] = (rng.random((N,), dtype=np.float32)).astype(np.float16) if in_dt == "float16" else \
            rng.integers(-8, 8, size=(N,), dtype=np.int8)
    return feeds, codes


def main():
    assert os.path.isdir(REF), "reference checkout not present"
    quant, deq, dense = install_tvm_stand_in()
    ref_utils = load_reference_utils()
    rng = np.random.default_rng(2024)
    arrays, meta = {}, []
    M, N, K = 3, 32, 256
    def both_models(run):
        """evaluate under C integer promotion (recorded) and under nominal-dtype arithmetic (compared)"""
        Interp.c_promotion = True
        got = run()
        Interp.c_promotion = False
        alt = run()
        Interp.c_promotion = True
        return got, alt

    for tag, kw in dequant_cases():
        feeds, codes = build_inputs(tag, kw, ref_utils, rng, M, N, K)

        def run():
            Interp.feeds, Interp.stages, Interp.const_overflow = feeds, {}, False
            func = deq.matmul_nt_dequantize_b(M, N, K, **kw)
            return func, Interp.stages["B_decode"], func.args[-1], Interp.const_overflow

        (func, bdec, out, ovf), (_, bdec_t, out_t, _) = both_models(run)
        arg_names = [t.name for t in func.args[:-1]]
        for name in arg_names:
            arrays[f"{tag}__{name}"] = np.asarray(feeds[name])
        arrays[f"{tag}__codes"] = codes.astype(np.uint8)
        arrays[f"{tag}__B_decode"] = bdec.data
        arrays[f"{tag}__out"] = out.data
        differs = not (np.array_equal(bdec.data, bdec_t.data, equal_nan=True) and np.array_equal(out.data, out_t.data, equal_nan=True))
        meta.append({"tag": tag, "kind": "dequant", "M": M, "N": N, "K": K, "kwargs": kw, "args": arg_names,
                     "B_decode_dtype": bdec.dtype, "out_dtype": out.dtype, "const_overflow": ovf,
                     "typed_model_differs": differs})
    # dense definitions (matmul_impl.py:49-84): fp8 x fp8 (c5), int8 x int8, fp16 x fp16
    import torch
    for tag, in_dt, acc, out_dt, bias in (("dense_e4m3", "e4m3_float8", "float32", "float16", False),
                                          ("dense_e4m3_f32", "e4m3_float8", "float32", "float32", False),
                                          ("dense_e5m2", "e5m2_float8", "float32", "float32", False),
                                          ("dense_i8", "int8", "int32", "int32", False),
                                          ("dense_i8_bias", "int8", "int32", "int32", True),
                                          ("dense_f16_bias", "float16", "float32", "float16", True)):
        if in_dt in FP8:
            tdt = torch.float8_e4m3fn if in_dt == "e4m3_float8" else torch.float8_e5m2
            # every byte value of the format appears in A and W (NaN / inf encodings excluded)
            allb = np.arange(256, dtype=np.uint8)
            vals = _fp8_to_f32(allb, in_dt)
            ok = allb[np.isfinite(vals)]
            if out_dt == "float16":
                ok = allb[np.isfinite(vals) & (np.abs(vals) <= 2.0)]      # keep the sums inside the float16 range
            A = rng.choice(ok, size=(M, K)).astype(np.uint8)
            W = rng.choice(ok, size=(N, K)).astype(np.uint8)
            W[0, : len(ok)] = ok
            A[0, : len(ok)] = ok[::-1]
            feeds = {"A": A, "B": W}
            del tdt
        elif in_dt == "int8":
            feeds = {"A": rng.integers(-128, 128, size=(M, K), dtype=np.int8), "B": rng.integers(-128, 128, size=(N, K), dtype=np.int8)}
        else:
            feeds = {"A": (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16),
                     "B": (rng.random((N, K), dtype=np.float32) - 0.5).astype(np.float16)}
        if bias:
            feeds["Bias"] = rng.integers(-8, 8, size=(N,), dtype=np.int8) if in_dt == "int8" else \
                rng.random((N,), dtype=np.float32).astype(np.float16)
        def run_dense():
            Interp.feeds, Interp.stages = feeds, {}
            f = dense.matmul_nt(M, N, K, in_dtype=in_dt, out_dtype=out_dt, accum_dtype=acc, with_bias=bias)
            return f, f.args[-1]

        (func, out), (_, out_t) = both_models(run_dense)
        assert np.array_equal(out.data, out_t.data, equal_nan=True)
        arg_names = [t.name for t in func.args[:-1]]
        for name in arg_names:
            arrays[f"{tag}__{name}"] = np.asarray(feeds[name])
        arrays[f"{tag}__out"] = out.data
        meta.append({"tag": tag, "kind": "dense", "M": M, "N": N, "K": K, "in_dtype": in_dt, "accum_dtype": acc,
                     "out_dtype": out_dt, "with_bias": bias, "args": arg_names})
    # the per-element decoders on their own, exhaustively: every (bit, pos) of every storage byte
    allbytes = np.arange(256, dtype=np.uint8).view(np.int8)
    typed_differs = []

    def record(key, fn, bits_view=False):
        got, alt = both_models(lambda: fn().nominal())
        if bits_view:
            got, alt = got.view(np.uint16), alt.view(np.uint16)
        arrays[key] = got
        if not np.array_equal(got, alt, equal_nan=True):
            typed_differs.append(key)

    for bit in (1, 2, 4):
        n = 8 // bit
        val = lambda: E(np.repeat(allbytes, n), "int8")                                   # noqa: E731
        pos = lambda: E(np.tile(np.arange(n, dtype=np.int32), 256), "int32")              # noqa: E731
        record(f"dec_unsigned_b{bit}_f16", lambda: quant._tir_packed_to_unsigned_convert("int", 8)(bit, val(), pos(), "float16"))
        record(f"dec_signed_b{bit}_f16", lambda: quant._tir_packed_to_signed_convert("int", 8)(bit, val(), pos(), "float16"))
        record(f"dec_signed_b{bit}_i8", lambda: quant._tir_packed_to_signed_convert("int", 8)(bit, val(), pos(), "int8"))
        record(f"dec_int2int_b{bit}_f16", lambda: quant._tir_packed_int_to_int_convert("int", 8)(bit, val(), pos(), "float16"))
        record(f"dec_int2int_b{bit}_i8", lambda: quant._tir_packed_int_to_int_convert("int", 8)(bit, val(), pos(), "int8"))
        for z in (0, 1, (1 << bit) - 1):
            zero = lambda: E(np.full(256 * n, z, dtype=np.int8), "int8")                  # noqa: E731
            record(f"dec_withzeros_b{bit}_z{z}_f16",
                   lambda: quant._tir_packed_to_unsigned_convert_with_zeros("int", 8)(bit, val(), pos(), zero(), "float16"))
    val8 = lambda: E(allbytes, "int8")                                                    # noqa: E731
    pos0 = lambda: E(np.zeros(256, dtype=np.int32), "int32")                              # noqa: E731
    for z in (0, 1, 127, 128, 200, 255):
        zero = lambda: E(np.full(256, z, dtype=np.uint8).view(np.int8), "int8")           # noqa: E731
        record(f"dec_withzeros_b8_z{z}_f16",
               lambda: quant._tir_packed_to_unsigned_convert_with_zeros("int", 8)(8, val8(), pos0(), zero(), "float16"))
    record("dec_fp4_f16", lambda: quant._tir_packed_to_fp4_to_f16("int", 8)(
        4, E(np.repeat(allbytes, 2), "int8"), E(np.tile(np.arange(2, dtype=np.int32), 256), "int32"), "float16"))
    record("dec_e4m3_f16_bits", lambda: quant._tir_u8_to_f8_e4m3_to_f16(8, val8(), "float16"), bits_view=True)
    record("dec_e4m3_naive_f16_bits", lambda: quant._tir_u8_to_f8_e4m3_to_f16_naive(8, val8(), "float16"), bits_view=True)
    record("dec_e5m2_f16_bits", lambda: quant._tir_u8_to_f8_e5m2_to_f16(8, E(allbytes.view(np.uint8), "uint8"), "float16"), bits_view=True)
    np.savez_compressed(OUT_NPZ, **arrays)
    with open(OUT_JSON, "w") as f:
        json.dump({"generator": "oracle/gen_te_golden.py", "integer_model": "C promotion (TVM CodeGenC); typed_model_differs marks "
                   "what an evaluation in nominal TIR dtypes would change", "cases": meta,
                   "decoders": sorted(k for k in arrays if k.startswith("dec_")), "decoders_typed_model_differs": typed_differs},
                  f, indent=1)
    print(f"wrote {len(meta)} cases, {len(arrays)} arrays -> {os.path.relpath(OUT_NPZ)} "
          f"({os.path.getsize(OUT_NPZ) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
