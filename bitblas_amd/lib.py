"""ctypes binding of libwqaa_hip.so - the thin `cdll` shim between the Python operator API and
the hand-written HIP kernels.

Reference counterpart: `Operator.update_runtime_module(libpath=...)` -> `ctypes.CDLL(libpath);
lib.init()` and `Operator._forward_from_prebuild_lib` (bitblas/ops/operator.py:458-485), where each
config owns a JIT-compiled `.so` exporting `init()` / `call()`.  Here one prebuilt library serves every
config; `BoundLib` gives each operator an object with the same `init()` / `call(*ptrs, [m], stream)`
surface so code that drives `matmul.lib.call(...)` directly (e.g. `Linear.forward`,
bitblas/module/__init__.py:267-289) keeps working.

The library is mandatory: if it is missing or fails to load we raise - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwqaa_hip.so")

# enums of include/wqaa.h
F16, BF16, F32, I8, I32, E4M3, E5M2, I4 = range(8)
W_UINT, W_INT, W_NF, W_FP4, W_E4M3, W_E5M2, W_NATIVE = range(7)
Z_NONE, Z_ORIGINAL, Z_RESCALE, Z_QUANTIZED = range(4)
LAYOUT_PLAIN, LAYOUT_LOP3 = 0, 1
OK, ERR_BAD_DESC, ERR_UNSUPPORTED, ERR_LAUNCH, ERR_NO_DEVICE = range(5)
EPI_QUANTIZE_INPUT = 1
EPI_ADD_RESIDUAL = 2
EPI_RMSNORM_INPUT = 4

DTYPE_CODE = {
    "float16": F16, "bfloat16": BF16, "float32": F32, "int8": I8, "int32": I32,
    "e4m3_float8": E4M3, "e5m2_float8": E5M2,
    "int4": I4,   # activations only: two's-complement nibbles, two per byte (A is (M, K/2) int8)
}
WFORMAT_CODE = {"uint": W_UINT, "int": W_INT, "nf": W_NF, "fp": W_FP4, "fp_e4m3": W_E4M3,
                "fp_e5m2": W_E5M2}
ZEROS_CODE = {"original": Z_ORIGINAL, "rescale": Z_RESCALE, "quantized": Z_QUANTIZED}

EXPORTED_SYMBOLS = (
    "init", "wqaa_abi_version", "wqaa_device_count", "wqaa_matmul", "wqaa_matmul_timed",
    "wqaa_matmul_ex", "wqaa_matmul_opts", "wqaa_workspace_bytes", "wqaa_matmul_group", "wqaa_matmul_group_ex", "wqaa_group_plan", "wqaa_matmul_gate_up", "wqaa_gate_up_plan", "wqaa_matmul_chain", "wqaa_chain_plan", "wqaa_dequantize", "wqaa_tune", "wqaa_act_quant_int8", "wqaa_select", "wqaa_select_ex", "wqaa_pack_weight", "wqaa_unpack_weight", "wqaa_relayout_weight", "wqaa_debug_decode", "wqaa_debug_row_blocks", "wqaa_debug_tile_of_block",
    "wqaa_peer_alloc", "wqaa_peer_free", "wqaa_peer_export", "wqaa_peer_open", "wqaa_peer_close", "wqaa_peer_exchange",
    "wqaa_last_error", "wqaa_last_error_string",
)


class MatmulDesc(ctypes.Structure):
    """struct wqaa_matmul_desc (include/wqaa.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("a_dtype", ctypes.c_int32), ("w_format", ctypes.c_int32), ("w_bits", ctypes.c_int32),
        ("out_dtype", ctypes.c_int32), ("group_size", ctypes.c_int32),
        ("with_scaling", ctypes.c_int32), ("zeros_mode", ctypes.c_int32),
        ("with_bias", ctypes.c_int32), ("w_layout", ctypes.c_int32),
        ("strict_reference", ctypes.c_int32), ("k_split_hint", ctypes.c_int32), ("two_pass_min_m", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 1),
    ]


class Plan(ctypes.Structure):
    """struct wqaa_plan (include/wqaa.h)."""
    _fields_ = [
        ("kernel_family", ctypes.c_int32), ("block_m", ctypes.c_int32), ("block_n", ctypes.c_int32),
        ("block_k", ctypes.c_int32), ("threads", ctypes.c_int32), ("grid", ctypes.c_int32),
        ("rows_per_wave", ctypes.c_int32), ("batch_tile", ctypes.c_int32),
        ("pipeline_depth", ctypes.c_int32), ("split_k", ctypes.c_int32),
        ("lds_bytes", ctypes.c_int32), ("name", ctypes.c_char * 96),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "name"}
        d["name"] = self.name.decode()
        return d


class Epilogue(ctypes.Structure):
    """struct wqaa_epilogue (include/wqaa.h): fused `out / si / sw -> half` of BitNet-style callers."""
    _fields_ = [("struct_size", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("row_scale", ctypes.c_void_p), ("tensor_scale", ctypes.c_float),
                ("reserved2", ctypes.c_int32), ("residual", ctypes.c_void_p), ("norm_weight", ctypes.c_void_p),
                ("norm_eps", ctypes.c_float), ("reserved3", ctypes.c_int32)]


def norm_epilogue(weight_ptr, eps) -> Epilogue:
    """wqaa_epilogue with WQAA_EPI_RMSNORM_INPUT: the RMSNorm (weight (K,) float16, variance_epsilon) in front of the operator"""
    epi = Epilogue()
    epi.struct_size = ctypes.sizeof(Epilogue)
    epi.flags = EPI_RMSNORM_INPUT
    epi.tensor_scale = 1.0
    epi.norm_weight = weight_ptr
    epi.norm_eps = float(eps)
    return epi


class CallOpts(ctypes.Structure):
    """struct wqaa_call_opts (include/wqaa.h): caller-owned split-K workspace (+ optional fused epilogue)."""
    _fields_ = [("struct_size", ctypes.c_int32), ("flags", ctypes.c_int32), ("workspace", ctypes.c_void_p),
                ("workspace_bytes", ctypes.c_uint64), ("epilogue", ctypes.POINTER(Epilogue))]


PEER_MAX = 16
PEER_HANDLE_BYTES = 64


class PeerExchangeDesc(ctypes.Structure):
    """struct wqaa_peer_exchange_desc (include/wqaa.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("bytes", ctypes.c_size_t), ("world", ctypes.c_int32), ("rank", ctypes.c_int32),
                ("step", ctypes.c_uint32), ("timeout_ms", ctypes.c_uint32),
                ("dst", ctypes.c_void_p * PEER_MAX), ("post", ctypes.c_void_p * PEER_MAX),
                ("flags", ctypes.c_void_p), ("status", ctypes.c_void_p)]


class WqaaError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libwqaa_hip error {code}: {message}")
        self.code = code


_lib = None

# WQAA_PLAN_LOG=<file>: every (operator, row count) the Python layer launches appends its plan name once - what a parity run
# actually exercised (tools/member_coverage.py compares it with what the selector can reach).  Hooked where a row count is
# first seen (`workspace_need`), so the launch path itself pays nothing when the variable is unset.
_PLAN_LOG = os.environ.get("WQAA_PLAN_LOG") or None
_plan_logged = set()


def _log_plan(desc, m, tag=""):
    key = (bytes(desc), int(m), tag)
    if key in _plan_logged:
        return
    _plan_logged.add(key)
    try:
        name = select(desc, m)["name"] + tag
    except WqaaError as exc:
        name = f"refused({exc})"
    with open(_PLAN_LOG, "a") as f:
        f.write(f"{int(m)}\t{name}\n")
_lib_lock = threading.Lock()


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the kernel library once, declare prototypes, call init()."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get("WQAA_LIBRARY", LIB_PATH)
        if not os.path.exists(p):
            raise ImportError(
                f"{p} not found: build the HIP kernel library first "
                f"(`python -m bitblas_amd.build`); bitblas_amd has no CPU fallback")
        lib = ctypes.CDLL(p)
        vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        dp = ctypes.POINTER(MatmulDesc)
        lib.init.restype = None
        lib.init.argtypes = []
        lib.wqaa_abi_version.restype = ci
        lib.wqaa_device_count.restype = ci
        lib.wqaa_matmul.restype = ci
        lib.wqaa_matmul.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, ci, vp]
        lib.wqaa_matmul_timed.restype = ci
        lib.wqaa_matmul_timed.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp]
        lib.wqaa_matmul_ex.restype = ci
        lib.wqaa_matmul_ex.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, ci, vp, ctypes.POINTER(Epilogue)]
        lib.wqaa_matmul_opts.restype = ci
        lib.wqaa_matmul_opts.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, ci, vp, ctypes.POINTER(CallOpts)]
        lib.wqaa_workspace_bytes.restype = ctypes.c_uint64
        lib.wqaa_workspace_bytes.argtypes = [dp, ci]
        lib.wqaa_act_quant_int8.restype = ci
        lib.wqaa_act_quant_int8.argtypes = [vp, i64, ci, vp, vp, vp]
        lib.wqaa_tune.restype = ci
        lib.wqaa_tune.argtypes = [dp, ci, vp]
        lib.wqaa_dequantize.restype = ci
        lib.wqaa_dequantize.argtypes = [dp, vp, vp, vp, vp, vp, vp]
        lib.wqaa_select.restype = ci
        lib.wqaa_select.argtypes = [dp, ci, ctypes.POINTER(Plan)]
        lib.wqaa_select_ex.restype = ci
        lib.wqaa_select_ex.argtypes = [dp, ci, ci, ctypes.POINTER(Plan)]
        lib.wqaa_pack_weight.restype = ci
        lib.wqaa_pack_weight.argtypes = [vp, i64, i64, ci, ci, ci, vp]
        lib.wqaa_unpack_weight.restype = ci
        lib.wqaa_unpack_weight.argtypes = [vp, i64, i64, ci, ci, ci, vp]
        lib.wqaa_relayout_weight.restype = ci
        lib.wqaa_relayout_weight.argtypes = [vp, i64, i64, ci, ci, ci, ci, vp]
        lib.wqaa_debug_decode.restype = ci
        lib.wqaa_debug_decode.argtypes = [vp, i64, ci, ci, ci, ci, ci, vp, vp, vp]
        lib.wqaa_peer_alloc.restype = ci
        lib.wqaa_peer_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
        lib.wqaa_peer_free.restype = ci
        lib.wqaa_peer_free.argtypes = [vp]
        lib.wqaa_peer_export.restype = ci
        lib.wqaa_peer_export.argtypes = [vp, vp]
        lib.wqaa_peer_open.restype = ci
        lib.wqaa_peer_open.argtypes = [vp, ctypes.POINTER(vp)]
        lib.wqaa_peer_close.restype = ci
        lib.wqaa_peer_close.argtypes = [vp]
        lib.wqaa_peer_exchange.restype = ci
        lib.wqaa_peer_exchange.argtypes = [ctypes.POINTER(PeerExchangeDesc), vp]
        lib.wqaa_last_error.restype = ci
        lib.wqaa_last_error_string.restype = ctypes.c_char_p
        if lib.wqaa_abi_version() != 4:
            raise ImportError(f"{p}: ABI version {lib.wqaa_abi_version()} != 4 (rebuild: python -m bitblas_amd.build)")
        lib.init()
        if path is None:
            _lib = lib
        return lib


def current_stream_handle(device) -> int:
    """hipStream_t of torch's current stream on `device` as an integer.  `torch.cuda.current_stream(dev).cuda_stream` builds a
    Stream object per call (~1.5 us, a quarter of an M = 1 kernel); torch's own raw accessor is a tenth of that."""
    try:
        import torch
        idx = device.index
        if idx is None:
            idx = torch.cuda.current_device()
        return torch._C._cuda_getCurrentRawStream(idx)
    except AttributeError:       # a torch without the raw accessor
        import torch
        return torch.cuda.current_stream(device).cuda_stream


def check(status: int) -> None:
    if status != OK:
        lib = load_library()
        raise WqaaError(status, lib.wqaa_last_error_string().decode(errors="replace"))


def make_desc(*, N, K, a_dtype, w_format, w_bits, out_dtype, group_size=-1, with_scaling=False,
              zeros_mode=Z_NONE, with_bias=False, w_layout=LAYOUT_PLAIN, strict_reference=False, k_split_hint=0, two_pass_min_m=0) -> MatmulDesc:
    d = MatmulDesc()
    d.struct_size = ctypes.sizeof(MatmulDesc)
    d.N, d.K = int(N), int(K)
    d.a_dtype, d.w_format, d.w_bits, d.out_dtype = int(a_dtype), int(w_format), int(w_bits), int(out_dtype)
    d.group_size = int(group_size)
    d.with_scaling = int(bool(with_scaling))
    d.zeros_mode = int(zeros_mode)
    d.with_bias = int(bool(with_bias))
    d.w_layout = int(w_layout)
    d.strict_reference = int(bool(strict_reference))
    d.k_split_hint = max(0, int(k_split_hint or 0))
    d.two_pass_min_m = max(0, int(two_pass_min_m or 0))
    return d


def select(desc: MatmulDesc, m: int) -> dict:
    lib = load_library()
    plan = Plan()
    check(lib.wqaa_select(ctypes.byref(desc), int(m), ctypes.byref(plan)))
    return plan.as_dict()


def _ptr(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, ctypes.c_void_p):
        return x.value
    if isinstance(x, int):
        return x
    return x.data_ptr()  # torch.Tensor


# Large scratch (> 16 MiB: the two-pass member's B_decode, N*K*sizeof(A_dtype)) is shared by every operator of a
# (stream, device): one buffer, grown geometrically; a buffer that is replaced is kept alive (a captured hipGraph may
# still point at it) - the ownership rules of the C library's own pool (csrc/wqaa_gemm.hip), held by the caller.
_SHARED_WS_MIN = 16 << 20
_shared_ws = {}
_shared_ws_retired = []


def shared_workspace(stream, device, need: int):
    key = (stream, str(device))
    ws = _shared_ws.get(key)
    if ws is None or ws.numel() < need:
        import torch
        if ws is not None:
            _shared_ws_retired.append(ws)
            need = max(need, 2 * ws.numel())
        ws = _shared_ws[key] = torch.empty(need, dtype=torch.uint8, device=device if device is not None else "cuda")
    return ws


class BoundLib:
    """Per-operator view of the library with the reference's `init()` / `call()` convention.

    call(A, B, [LUT], [Scale], [Zeros|QZeros], [Bias], C, [m], stream)
      - pointer arguments: ctypes.c_void_p, raw ints or torch tensors
      - `m` is present iff the operator was built for a dynamic M range
      - `stream`: ctypes.c_void_p / int (hipStream_t)
    (argument order: tirscript/matmul_dequantize_impl.py:465-478; caller: ops/operator.py:458-463)
    """

    def __init__(self, desc: MatmulDesc, *, has_lut: bool, dynamic_m: bool, static_m: int = 1,
                 default_lut=None):
        self._lib = load_library()
        self.desc = desc
        self._desc_ref = ctypes.byref(desc)
        self.has_lut = has_lut
        self.has_scale = bool(desc.with_scaling)
        self.has_zeros = desc.zeros_mode != Z_NONE
        self.has_bias = bool(desc.with_bias)
        self.dynamic_m = dynamic_m
        self.static_m = static_m
        self.default_lut = default_lut
        self._n_opt = int(self.has_scale) + int(self.has_zeros) + int(self.has_bias)
        self._fn = self._lib.wqaa_matmul
        self._ws_need = {}     # m -> bytes of scratch the selected member needs
        self._ws = {}          # (stream, device) -> torch uint8 tensor

    def init(self):
        self._lib.init()

    def call(self, *args):
        args = list(args)
        stream = _ptr(args.pop())
        m = int(args.pop()) if self.dynamic_m else self.static_m
        ptrs = [_ptr(a) for a in args]
        want = 3 + self._n_opt + int(self.has_lut)
        lut = None
        if self.has_lut:
            if len(ptrs) == want:
                lut = ptrs.pop(2)
            elif len(ptrs) == want - 1:
                # `Linear.forward` never passes the LUT (module/__init__.py:138-153)
                lut = _ptr(self.default_lut)
            else:
                raise TypeError(f"call() expected {want} pointer arguments, got {len(ptrs)}")
        elif len(ptrs) != want:
            raise TypeError(f"call() expected {want} pointer arguments, got {len(ptrs)}")
        it = iter(ptrs)
        A, B = next(it), next(it)
        scale = next(it) if self.has_scale else None
        zeros = next(it) if self.has_zeros else None
        bias = next(it) if self.has_bias else None
        C = next(it)
        self.run(A, B, lut, scale, zeros, bias, C, m, stream)

    def run(self, A, B, lut, scale, zeros, bias, C, m, stream, device=None):
        """Fast path used by Matmul.forward: raw integer pointers, no list juggling.

        Members that need scratch (split-K partial sums, `wqaa_workspace_bytes`) get a CALLER-OWNED workspace here, the
        reference's ownership model (general_matmul/__init__.py:29, 456-457, 482): one torch tensor per (operator,
        stream), allocated by torch's caching allocator - which also works while the stream is being captured into a
        graph (the block then lives in the graph's private pool for the graph's lifetime; `torch.cuda.graph` captures
        on a side stream the library's own per-stream slab has never seen).  Two streams never share a workspace."""
        need = self.workspace_need(m)
        if need:
            if need > _SHARED_WS_MIN:
                # the two-pass member's B_decode scratch is N*K*sizeof(A_dtype): one buffer per (stream, device) for ALL
                # operators (launches on a stream are ordered; a per-operator copy would be a float16 copy of the model)
                ws = shared_workspace(stream, device, need)
                return self.run_ws(A, B, lut, scale, zeros, bias, C, m, stream, ws.data_ptr(), need)
            key = (stream, device)
            ws = self._ws.get(key)
            if ws is None or ws.numel() < need:
                import torch
                size = need
                if ws is not None:
                    # a dynamic-M operator met a taller batch: the old buffer is RETIRED, not freed - a hipGraph captured
                    # at the smaller row count still points at it (same rule as the C pool and `shared_workspace`);
                    # geometric growth bounds what accumulates
                    _shared_ws_retired.append(ws)
                    size = max(need, 2 * ws.numel())
                ws = self._ws[key] = torch.empty(size, dtype=torch.uint8, device=device if device is not None else "cuda")
            return self.run_ws(A, B, lut, scale, zeros, bias, C, m, stream, ws.data_ptr(), need)
        status = self._fn(self._desc_ref, A, B, lut, scale, zeros, bias, C, m, stream)
        if status != OK:
            check(status)

    def run_ws(self, A, B, lut, scale, zeros, bias, C, m, stream, workspace_ptr, workspace_bytes):
        """as `run`, with a caller-owned split-K workspace (wqaa_matmul_opts)"""
        opts = CallOpts()
        opts.struct_size = ctypes.sizeof(CallOpts)
        opts.workspace = workspace_ptr
        opts.workspace_bytes = int(workspace_bytes)
        status = self._lib.wqaa_matmul_opts(self._desc_ref, A, B, lut, scale, zeros, bias, C, m, stream, ctypes.byref(opts))
        if status != OK:
            check(status)

    def workspace_bytes(self, m: int) -> int:
        return int(self._lib.wqaa_workspace_bytes(self._desc_ref, int(m)))

    def workspace_need(self, m: int) -> int:
        """`workspace_bytes`, asked once per row count"""
        need = self._ws_need.get(m)
        if need is None:
            need = self._ws_need[m] = self.workspace_bytes(m)
            if _PLAN_LOG:
                _log_plan(self.desc, m)
        return need

    def run_timed(self, A, B, lut, scale, zeros, bias, C, m, stream, ev_start, ev_stop):
        status = self._lib.wqaa_matmul_timed(self._desc_ref, A, B, lut, scale, zeros, bias, C, m,
                                             stream, ev_start, ev_stop)
        if status != OK:
            check(status)

    def run_fused(self, A, B, bias, C, m, stream, row_scale_ptr, tensor_scale):
        """int8 path with the caller's `out / si / sw -> half (+bias)` folded into the epilogue."""
        if _PLAN_LOG:
            _log_plan(self.desc, m, "+epi")
        epi = Epilogue()
        epi.struct_size = ctypes.sizeof(Epilogue)
        epi.row_scale = row_scale_ptr
        epi.tensor_scale = float(tensor_scale)
        status = self._lib.wqaa_matmul_ex(self._desc_ref, A, B, None, None, None, bias, C, m, stream,
                                          ctypes.byref(epi))
        if status != OK:
            check(status)

    def run_fused_quant(self, X, B, bias, C, m, stream, tensor_scale):
        """BitNet layer in one launch (m <= 4): X is the float16 input; the kernel applies activation_quant
        itself (WQAA_EPI_QUANTIZE_INPUT) and folds `out / si / sw -> half (+bias)` into its epilogue."""
        if _PLAN_LOG:
            _log_plan(self.desc, m, "+epi_quant")
        epi = Epilogue()
        epi.struct_size = ctypes.sizeof(Epilogue)
        epi.flags = EPI_QUANTIZE_INPUT
        epi.row_scale = None
        epi.tensor_scale = float(tensor_scale)
        status = self._lib.wqaa_matmul_ex(self._desc_ref, X, B, None, None, None, bias, C, m, stream,
                                          ctypes.byref(epi))
        if status != OK:
            check(status)

    def run_residual(self, A, B, scale, zeros, bias, C, m, stream, residual=None, norm=None):
        """float16 decode path with one of the caller's elementwise ops folded in (wqaa_matmul_ex): `residual` (pointer) -
        C = residual + matmul(...) (WQAA_EPI_ADD_RESIDUAL); `norm` = (weight pointer, eps) - A is the hidden state in front of
        the layer's RMSNorm (WQAA_EPI_RMSNORM_INPUT).  Raises WqaaError(UNSUPPORTED) where no exact-product GEMV member exists."""
        if _PLAN_LOG:
            _log_plan(self.desc, m, "+norm" if norm else "+residual")
        epi = norm_epilogue(*norm) if norm else Epilogue()
        epi.struct_size = ctypes.sizeof(Epilogue)
        epi.flags = EPI_RMSNORM_INPUT if norm else EPI_ADD_RESIDUAL
        epi.tensor_scale = 1.0
        epi.residual = residual
        status = self._lib.wqaa_matmul_ex(self._desc_ref, A, B, None, scale, zeros, bias, C, m, stream, ctypes.byref(epi))
        if status != OK:
            check(status)

    def tune(self, m: int, stream) -> None:
        """`wqaa_tune`: time the vendor library's candidate algorithms behind (desc, m) on the device, keep the fastest"""
        check(self._lib.wqaa_tune(self._desc_ref, int(m), stream))
        self._ws_need.clear()

    def plan(self, m: int) -> dict:
        self._ws_need.clear()      # planning re-reads the tuning variables: the scratch a member needs may change with them
        return select(self.desc, m)

    def plan_ex(self, m: int, epilogue_flags: int = 0) -> dict:
        """the member `run_fused` / `run_residual` take (wqaa_select_ex): 0 = the caller's row / tensor scales alone"""
        self._ws_need.clear()
        plan = Plan()
        check(load_library().wqaa_select_ex(ctypes.byref(self.desc), int(m), int(epilogue_flags), ctypes.byref(plan)))
        return plan.as_dict()


def act_quant_int8(x, q, s, stream):
    """x: (rows, K) float16 cuda tensor -> q int8 (rows, K), s float32 (rows,) through the HIP quantiser."""
    lib = load_library()
    rows = x.numel() // x.shape[-1]
    check(lib.wqaa_act_quant_int8(x.data_ptr(), rows, int(x.shape[-1]), q.data_ptr(), s.data_ptr(), stream))


def pack_weight(codes, bits: int, layout: int, a_dtype_code: int):
    """numpy int8 (rows, cols) unsigned codes -> numpy int8 (rows, cols*bits/8) via the C packer."""
    import numpy as np
    lib = load_library()
    c = np.ascontiguousarray(codes, dtype=np.int8)
    rows, cols = c.shape
    out = np.empty((rows, cols * bits // 8), dtype=np.int8)
    check(lib.wqaa_pack_weight(c.ctypes.data, rows, cols, bits, layout, a_dtype_code, out.ctypes.data))
    return out


def unpack_weight(packed, cols: int, bits: int, layout: int, a_dtype_code: int):
    import numpy as np
    lib = load_library()
    p = np.ascontiguousarray(packed).view(np.int8)
    rows = p.shape[0]
    out = np.empty((rows, cols), dtype=np.int8)
    check(lib.wqaa_unpack_weight(p.ctypes.data, rows, cols, bits, layout, a_dtype_code, out.ctypes.data))
    return out


def relayout_weight(packed, bits: int, from_layout: int, to_layout: int, a_dtype_code: int):
    """numpy (rows, row_bytes) packed bytes of one layout -> the other (PLAIN -> LOP3 is the reference's LOP3Permutate
    stage, ops/lop3_permutate/lop3_permutate_impl.py:12-132), word by word in C."""
    import numpy as np
    lib = load_library()
    p = np.ascontiguousarray(packed).view(np.int8)
    rows, row_bytes = p.shape
    out = np.empty_like(p)
    check(lib.wqaa_relayout_weight(p.ctypes.data, rows, row_bytes, bits, from_layout, to_layout, a_dtype_code, out.ctypes.data))
    return out
