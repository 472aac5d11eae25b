"""`MatmulConfig` / `Matmul`: the operator API of microsoft/BitBLAS on top of the gfx950 kernels.

Mirrors `bitblas/ops/general_matmul/__init__.py` (config :58-237, kernel names :240-318, operator
:321-841) and the slice of `bitblas/ops/operator.py` (:94-133, :458-485, :529-556) that the hot
path touches.  What is gone: TVM/TileLang lowering, the roller tuner, per-config JIT libraries.
What replaces them: `bitblas_amd.lib.BoundLib` (one prebuilt HIP library) and the C++ tile
selector behind `wqaa_select`.

Numerics contract (same as the TE definition, tirscript/matmul_dequantize_impl.py:339-499):
weights are dequantised in A_dtype, products are accumulated in fp32 (int32 for int8
activations) - the reference's default `accum_dtype="float16"` is accepted and honoured *at least*
as accurately - the result is cast to out_dtype and the bias is added after the cast.
"""
from __future__ import annotations

import logging
import operator as _operator
import re
from dataclasses import dataclass
from enum import IntEnum
from functools import reduce
from typing import Any, List, Literal, Optional, Tuple, Union

import numpy as np
import torch

from . import lib as _lib
from .target import auto_detect_nvidia_target, get_arch

logger = logging.getLogger(__name__)

WORKSPACE_SIZE = 1024 * 1024 * 256


class OptimizeStrategy(IntEnum):
    """bitblas/base/operator_common.py:7-15"""
    SingleBatchDecodeOnly = 0
    ContigousBatching = 1

    def is_single_batch_decode_only(self):
        return self is OptimizeStrategy.SingleBatchDecodeOnly

    def is_contigous_batching(self):
        return self is OptimizeStrategy.ContigousBatching


class TransformKind(IntEnum):
    """bitblas/base/operator_common.py:18-34"""
    NonTransform = 0
    InterWarpTransform = 1
    IntraWarpTransform = 2
    LDMatrixTransform = 3

    def is_non_transform(self):
        return self is TransformKind.NonTransform

    def is_inter_warp_transform(self):
        return self is TransformKind.InterWarpTransform

    def is_intra_warp_transform(self):
        return self is TransformKind.IntraWarpTransform

    def is_ld_matrix_transform(self):
        return self is TransformKind.LDMatrixTransform


NATIVE_COMPUTE_PATTERNS = frozenset({
    ("float64", "float64"), ("float32", "float32"), ("float16", "float16"),
    ("bfloat16", "bfloat16"), ("int8", "int8"), ("uint8", "uint8"), ("int4", "int4"),
    ("uint4", "uint4"), ("e4m3_float8", "e4m3_float8"), ("e4m3_float8", "e5m2_float8"),
    ("e5m2_float8", "e4m3_float8"), ("e5m2_float8", "e5m2_float8"),
})


def is_native_compute(A_dtype: str, W_dtype: str) -> bool:
    """general_matmul/__init__.py:33-51"""
    return (A_dtype, W_dtype) in NATIVE_COMPUTE_PATTERNS


@dataclass(frozen=True)
class OperatorConfig:
    """Typing root for operator configs (ops/operator.py:42-46)."""


_SAME_STORAGE_DTYPES = ("float16", "bfloat16", "int8", "e4m3_float8", "e5m2_float8")
_MICRO = 16


@dataclass(frozen=True)
class MatmulConfig(OperatorConfig):
    """Frozen, hashable description of one matmul; `repr()` is the operator-cache key.

    Field names, order, defaults and the post-init legalisation reproduce
    general_matmul/__init__.py:58-237 (including the upstream spelling `optimize_stratety`), so a
    config built here hashes to the same database key as upstream.
    """
    M: Union[int, Tuple[int]] = None
    N: Optional[int] = None
    K: Optional[int] = None
    A_dtype: str = "float16"
    W_dtype: str = A_dtype
    out_dtype: str = "float16"
    accum_dtype: str = "float16"
    layout: Literal["nn", "nt", "tn", "tt"] = "nt"
    with_bias: bool = False
    group_size: int = -1
    with_scaling: bool = False
    with_zeros: bool = False
    # original : (w - zero) * scale | rescale: w * scale - zero | quantized: (w - dq(qzero)) * scale
    zeros_mode: Literal["original", "rescale", "quantized"] = "original"
    storage_dtype: str = "int8"
    fast_decoding: Optional[bool] = None
    propagate_a: Optional[TransformKind] = None
    propagate_b: Optional[TransformKind] = None
    optimize_stratety: Union[int, OptimizeStrategy] = OptimizeStrategy.SingleBatchDecodeOnly

    # -- helpers -------------------------------------------------------------------------------
    def _set(self, name, value):
        object.__setattr__(self, name, value)

    @staticmethod
    def _as_transform_kind(value):
        if isinstance(value, bool):
            return TransformKind.LDMatrixTransform if value else TransformKind.NonTransform
        if isinstance(value, int):
            return TransformKind(value)
        return value

    def _default_fast_decoding(self) -> bool:
        """True iff W is a sub-byte integer type decoded into a different A type (:163-184)."""
        blockers = (
            "int" not in self.W_dtype,
            self.W_dtype == self.A_dtype,
            self.W_dtype in ("int8", "uint8"),
            self.W_dtype in ("int4", "uint4") and self.A_dtype == "int8",
            self.A_dtype == "bfloat16",
        )
        return not any(blockers)

    def _legalize_propagation(self, user_a, user_b):
        """:113-154 - kept for config/hash parity; gfx950 kernels never need a ladder layout."""
        m_is_range = isinstance(self.M, tuple)
        if user_b is not None and user_b == TransformKind.NonTransform:
            pa = TransformKind.NonTransform
        elif isinstance(self.M, int) and self.M % _MICRO == 0 and self.K % _MICRO == 0:
            pa = TransformKind.IntraWarpTransform
        else:
            pa = TransformKind.NonTransform
        if self.M == 1 or self.N % _MICRO != 0 or self.K % _MICRO != 0 or m_is_range:
            pa = pb = TransformKind.NonTransform
        else:
            pb = TransformKind.LDMatrixTransform
        if user_a is not None:
            pa = user_a
        if user_b is not None:
            pb = user_b
        if self.optimize_stratety == OptimizeStrategy.ContigousBatching and (
                self.M != 1 or (m_is_range and 1 not in self.M)):
            pb = TransformKind.LDMatrixTransform
        if self.A_dtype in ("e4m3_float8", "e5m2_float8", "bfloat16"):
            pa = pb = TransformKind.NonTransform
        if self.A_dtype in ("int4", "uint4"):
            pa = TransformKind.NonTransform
            if pb == TransformKind.IntraWarpTransform:
                pb = TransformKind.LDMatrixTransform
        self._set("propagate_a", pa)
        self._set("propagate_b", pb)

    def __post_init__(self):
        if self.M is None:
            single = self.optimize_stratety == OptimizeStrategy.SingleBatchDecodeOnly
            self._set("M", [1, 16, 32, 64, 128, 256, 512, 1024] if single else
                      [16, 32, 64, 128, 256, 512, 1024])
        if self.N is None:
            raise ValueError("N should be specified currently.")
        if self.K is None:
            raise ValueError("K should be specified currently.")
        if isinstance(self.M, list):
            self._set("M", tuple(self.M))
        user_a = self._as_transform_kind(self.propagate_a)
        user_b = self._as_transform_kind(self.propagate_b)
        self._set("propagate_a", user_a)
        self._set("propagate_b", user_b)
        if isinstance(self.optimize_stratety, int):
            self._set("optimize_stratety", OptimizeStrategy(self.optimize_stratety))
        self._legalize_propagation(user_a, user_b)
        if self.zeros_mode is None:
            self._set("zeros_mode", "original")
        if self.fast_decoding is None:
            self._set("fast_decoding", self._default_fast_decoding())
        if self.with_bias is None:
            self._set("with_bias", False)
        if self.group_size is None:
            self._set("group_size", -1)
        if self.with_scaling is None:
            self._set("with_scaling", False)
        if self.with_zeros is None:
            self._set("with_zeros", False)
        if self.A_dtype == self.W_dtype and self.W_dtype in _SAME_STORAGE_DTYPES:
            self._set("storage_dtype", self.W_dtype)


class BaseKernelNameGenerator:
    """ops/operator.py:49-70"""

    def __init__(self, config: OperatorConfig):
        assert self.is_valid_config(config), f"Invalid config for {type(self).__name__}: {config}"
        self.config = config

    def is_valid_config(self, config) -> bool:  # pragma: no cover - interface
        raise NotImplementedError

    def generate(self, hint=None) -> str:  # pragma: no cover - interface
        raise NotImplementedError

    @staticmethod
    def is_valid(kernel_name: str) -> bool:
        return bool(kernel_name.isidentifier() and re.match(r"^[A-Za-z_][A-Za-z0-9_]*$", kernel_name))


class MatmulKernelNameGenerator(BaseKernelNameGenerator):
    """`matmul_[m{M}]n{N}k{K}_{A}x{W}_{hint}` (general_matmul/__init__.py:240-318).

    `hint` is either None ("default"), a `wqaa_select` plan dict (our tile selector's answer), or
    any object with the reference Hint's attributes.
    """
    KERNEL_PREFIX = "matmul"

    @staticmethod
    def simplify_dtype(dtype: str) -> str:
        fixed = {"float32": "f32", "float16": "f16", "bfloat16": "bf16"}
        if dtype in fixed:
            return fixed[dtype]
        if dtype.startswith("int"):
            return "i" + dtype[3:]
        if dtype.startswith("uint"):
            return "u" + dtype[4:]
        return dtype

    @staticmethod
    def serialize_hint(hint=None) -> str:
        if hint is None:
            return "default"
        if isinstance(hint, dict):  # a wqaa_plan
            if hint.get("kernel_family") == 2:
                name = f"tcx{hint['block_m']}x{hint['block_n']}x{hint['block_k']}"
                if hint.get("split_k", 1) > 1:
                    name += f"xr{hint['split_k']}"
                if hint.get("pipeline_depth", 1) > 1:
                    name += f"xp{hint['pipeline_depth']}"
                return name
            return "simt"
        if getattr(hint, "use_tc", False):
            bm, bn = hint.block
            wm, wn = hint.warp
            name = f"tcx{bm}x{bn}x{hint.rstep[-1]}w{wm}x{wn}"
            rk = getattr(hint, "block_reduction_depth", None)
            if rk is not None and rk > 1:
                name += f"xr{rk}"
            if getattr(hint, "pipeline_stage", 1) > 1:
                name += f"xp{hint.pipeline_stage}"
            return name
        return "simt"

    def generate(self, hint=None) -> str:
        cfg = self.config
        shape = f"n{cfg.N}k{cfg.K}"
        if isinstance(cfg.M, int):
            shape = f"m{cfg.M}" + shape
        prec = f"{self.simplify_dtype(cfg.A_dtype)}x{self.simplify_dtype(cfg.W_dtype)}"
        name = "_".join([self.KERNEL_PREFIX, shape, prec, self.serialize_hint(hint)])
        assert self.is_valid(name), "Kernel name invalid"
        return name

    def is_valid_config(self, config) -> bool:
        return isinstance(config, MatmulConfig)


_TORCH_DTYPE = {
    "float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32,
    "float64": torch.float64, "int8": torch.int8, "uint8": torch.uint8, "int32": torch.int32,
    "e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2,
    "int4": torch.int8, "uint4": torch.int8,   # sub-byte dtypes travel packed in int8 storage
}


def torch_dtype(name: str) -> torch.dtype:
    return _TORCH_DTYPE[name] if name in _TORCH_DTYPE else getattr(torch, name)


class _QuantCompress:
    """CPU bit-packing stage (reference: ops/quant_compress, TVM-llvm; here the C packer)."""

    def __init__(self, bits: int, a_code: int):
        self.bits, self.a_code = bits, a_code

    def forward(self, w: torch.Tensor) -> torch.Tensor:
        codes = w.detach().cpu().contiguous().to(torch.int8).numpy()
        return torch.from_numpy(_lib.pack_weight(codes, self.bits, _lib.LAYOUT_PLAIN, self.a_code))


class _Lop3Permutate:
    """CPU LOP3 interleave stage (reference: ops/lop3_permutate); keeps the checkpoint layout."""

    def __init__(self, bits: int, a_code: int):
        self.bits, self.a_code = bits, a_code

    def forward(self, packed: torch.Tensor) -> torch.Tensor:
        p = packed.detach().cpu().contiguous().view(torch.int8).numpy()
        return torch.from_numpy(_lib.relayout_weight(p, self.bits, _lib.LAYOUT_PLAIN, _lib.LAYOUT_LOP3, self.a_code))


class OPExecutorCPU:
    """Chain of CPU-side transforms (ops/operator.py:529-556)."""

    def __init__(self, operators: Optional[List] = None):
        self.operators = list(operators) if operators else []

    def append(self, op):
        self.operators.append(op)

    def is_none(self):
        return not self.operators

    def forward(self, weight):
        out = weight
        for op in self.operators:
            out = op.forward(out)
        return out

    __call__ = forward

    @property
    def size(self):
        return len(self.operators)


def rms_norm_reference(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """the reference's BitnetRMSNorm.forward (= LlamaRMSNorm; integration/BitNet/modeling_bitnet.py:99-104) as torch ops: what
    the fused launches compute in front of the operator, and what runs where they do not exist"""
    h = x.to(torch.float32)
    variance = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(variance + eps)
    return weight * h.to(x.dtype)


def check_norm(norm, A: torch.Tensor, K: int) -> None:
    """`norm` = (weight, eps): the kernels read K float16 weights through a raw pointer"""
    weight, eps = norm
    if weight.dtype != A.dtype or weight.numel() != K or weight.device != A.device or not weight.is_contiguous():
        raise ValueError(f"the norm weight must hold {K} contiguous {A.dtype} elements on A's device")
    if not float(eps) >= 0.0:
        raise ValueError("the norm's eps must be a non-negative number")


class Operator:
    """`bitblas.ops.Operator` (ops/operator.py:92-527): the type callers annotate and check against, and the calls they make
    on any operator - `op(*tensors)`, `hardware_aware_finetune`, `profile_latency`, `get_source`, `cleanup`.  Upstream's
    base class owns the TVM build / wrap / compile pipeline; with a prebuilt kernel library that part is empty, so the base
    keeps only the backend predicates and the call protocol.  Subclasses define `forward`."""

    backend = "tl"

    def is_tir_backend(self):
        return self.backend == "tir"

    def is_tilelang_backend(self):
        return self.backend == "tl"

    def forward(self, *args, **kwargs):  # pragma: no cover - interface
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def cleanup(self):
        pass


class Matmul(Operator):
    """Mixed-precision `C = A @ dq(W)^T (+bias)` on MI355X; API of `bitblas.Matmul` (:321-841).

    `strict_reference` (not in the reference's signature; default False): False lets the selector take the members that skip
    an intermediate rounding of the TE definition - the exact-product GEMV for M <= 2 (group scale applied to fp32 partial
    sums instead of rounding every dequantised weight to float16), the IEEE e4m3 decode - all inside the reference's 1e-3
    contract and closer to the real-valued product (tests/test_gemvx_gpu.py).  True pins the definition to the letter:
    per-element float16 rounding of B_decode (matmul_dequantize_impl.py:391-459), the e4m3 bit trick with 0 -> 2^-7
    (quantization.py:169-176), "uint8" weights through the signed storage type.  `Linear` and `install_as_bitblas()` callers
    get the default."""

    BITBLAS_TRICK_DTYPE_MAP = {
        "float64": ("fp", 64), "float32": ("fp", 32), "float16": ("fp", 16), "bfloat16": ("bf", 16),
        "int32": ("int", 32), "uint32": ("uint", 32), "int16": ("int", 16), "uint16": ("uint", 16),
        "int8": ("int", 8), "uint8": ("uint", 8), "int4": ("int", 4), "uint4": ("uint", 4),
        "int2": ("int", 2), "uint2": ("uint", 2), "int1": ("int", 1), "uint1": ("uint", 1),
        "nf4": ("nf", 4), "fp4_e2m1": ("fp", 4),
        "e4m3_float8": ("fp_e4m3", 8),  # torch.float8_e4m3fn
        "e5m2_float8": ("fp_e5m2", 8),
    }

    NF4_VALUES = (
        -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
        -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
        0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
        0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0,
    )

    def __init__(self, config: MatmulConfig, name: str = "matmul", target: Optional[str] = None,
                 enable_tuning: bool = True, from_database: bool = False, backend: str = "tl",
                 device: Optional[Union[str, torch.device]] = None, strict_reference: bool = False):
        if target is None:
            target = auto_detect_nvidia_target()
        assert config.A_dtype in self.BITBLAS_TRICK_DTYPE_MAP, f"Unsupported input dtype {config.A_dtype}"
        assert config.W_dtype in self.BITBLAS_TRICK_DTYPE_MAP, f"Unsupported weight dtype {config.W_dtype}"
        if config.layout != "nt":
            raise ValueError("Only the 'nt' layout (A row-major, W [N, K]) is implemented")
        self.name = name
        self.config = config
        self.target = target
        self.backend = backend
        self.arch = get_arch(target)
        if self.arch.platform != "CDNA":           # (get_arch refuses everything but hip / gfx950 already; ref :388-389 names cuda and hip)
            raise ValueError("bitblas_amd only supports the hip (gfx950) target")
        self.source_format, self.bit = self.BITBLAS_TRICK_DTYPE_MAP[config.W_dtype]
        if self.source_format == "int" and config.with_zeros:
            logger.warning("[BitBLAS][Warning] with_zeros is not supported for int source format "
                           "as int has a constant zeropoints already.")
        self.strict_reference = strict_reference
        self.device = torch.device(device) if device is not None else None
        self.kernel_name_generator = self.get_kernel_name_generator()
        self.dynamic_range = {"m": config.M} if isinstance(config.M, tuple) else None
        self.workspace = None
        self.lut = None
        self._lut_device = None
        self.torch_output_dtype = torch_dtype(config.out_dtype)

        a_code = _lib.DTYPE_CODE.get(config.A_dtype)
        out_code = _lib.DTYPE_CODE.get(config.out_dtype)
        if a_code is None or out_code is None:
            raise ValueError(f"A_dtype={config.A_dtype} / out_dtype={config.out_dtype} have no gfx950 kernel")
        native = is_native_compute(config.A_dtype, config.W_dtype)
        if native:
            w_format, w_bits = _lib.W_NATIVE, self.bit
            if config.A_dtype != config.W_dtype:  # mixed fp8 pair: W carries its own format
                w_format = _lib.WFORMAT_CODE[self.source_format]
        else:
            w_format, w_bits = _lib.WFORMAT_CODE[self.source_format], self.bit
        zeros_mode = _lib.ZEROS_CODE[config.zeros_mode] if config.with_zeros else _lib.Z_NONE
        self._a_code = a_code
        self._desc = _lib.make_desc(
            N=config.N, K=config.K, a_dtype=a_code, w_format=w_format, w_bits=w_bits,
            out_dtype=out_code, group_size=config.group_size, with_scaling=config.with_scaling,
            zeros_mode=zeros_mode, with_bias=config.with_bias,
            w_layout=_lib.LAYOUT_LOP3 if (config.fast_decoding and not native) else _lib.LAYOUT_PLAIN,
            strict_reference=strict_reference, k_split_hint=getattr(config, "k_split", 0))
        static_m = config.M if isinstance(config.M, int) else 1
        self.lib = _lib.BoundLib(self._desc, has_lut=self.source_format == "nf",
                                 dynamic_m=self.dynamic_range is not None, static_m=static_m)
        # fail at construction (not at the first forward) when no kernel covers this config
        probe_m = list(config.M) if isinstance(config.M, tuple) else [config.M]
        self.plans = {m: self.lib.plan(m) for m in probe_m}

        # operand sizes the kernels will read (checked on every forward: a short buffer is an out-of-bounds read)
        self._a_cols = config.K // 2 if config.A_dtype in ("int4", "uint4") else config.K
        self._a_torch_dtype = torch_dtype(config.A_dtype)
        self._w_bytes = config.N * config.K * self.bit // 8
        self.weight_compress = _QuantCompress(self.bit, a_code) if self.bit in (1, 2, 4) else None
        self.lop3_permutate = None
        if config.fast_decoding and not native:
            assert self.source_format in ("int", "uint")
            self.lop3_permutate = _Lop3Permutate(self.bit, a_code)
        self.ladder_permutate_a = None
        self.ladder_permutate_b = None
        self.input_executors = OPExecutorCPU()
        self.weight_executors = OPExecutorCPU(
            [op for op in (self.weight_compress, self.lop3_permutate) if op is not None])
        if enable_tuning:
            self.hardware_aware_finetune()

    # -- reference API surface -------------------------------------------------------------------
    def get_kernel_name_generator(self):
        return MatmulKernelNameGenerator(self.config)

    def hardware_aware_finetune(self, topk: int = 20, parallel_build: bool = True):
        """The reference runs the roller + profiler here and keeps the fastest candidate (ops/operator.py:262-293, 347-382).
        The static library's tile choice is `wqaa_select`'s table, and every member it can pick is one of this library's own
        kernels; there is nothing to measure by default, the call refreshes the recorded plans (so `Linear.warmup` keeps
        working).  Only with the vendor-library yardstick opted in (WQAA_DENSE_LIB=1, a plan-time switch) is one choice timed on
        the device: from which row count on the TWO-PASS member (B_decode to a scratch, then hipBLASLt's plain GEMM -
        `wqaa_matmul_desc.two_pass_min_m`) beats the fused MFMA member; the smallest M where it wins by > 3 % becomes the
        threshold."""
        cand = sorted(m for m in self.plans if isinstance(m, int) and m >= 256)
        native = self.W_dtype == self.A_dtype
        if cand and torch.cuda.is_available():
            # the vendor GEMM behind the operator (plain dense pairs / second pass of the two-pass member): candidates timed
            dev = self.device if isinstance(self.device, torch.device) else torch.device("cuda", torch.cuda.current_device())
            try:
                for m in cand:
                    self.lib.tune(m, _lib.current_stream_handle(dev))
            except (_lib.WqaaError, RuntimeError) as exc:
                logger.info("library tuning skipped: %s", exc)
        if cand and not native and not self.with_bias and torch.cuda.is_available() and self.A_dtype in ("float16", "bfloat16", "int8"):
            try:
                self._tune_two_pass(cand)
            except (_lib.WqaaError, RuntimeError) as exc:      # no member / no memory for the scratch: the fused members stay
                logger.info("two-pass tuning skipped: %s", exc)
                self._desc.two_pass_min_m = 0
        self.plans = {m: self.lib.plan(m) for m in self.plans}
        return self.plans

    def _tune_two_pass(self, cand, iters: int = 5):
        dev = self.device if isinstance(self.device, torch.device) else torch.device("cuda", torch.cuda.current_device())
        a_dt = torch_dtype(self.A_dtype)
        g = self.K if self.group_size in (-1, None) else self.group_size
        W = torch.randint(-128, 127, self.retrieve_weight_shape(), device=dev, dtype=torch.int8)
        scale = (torch.rand(self.N, self.K // g, device=dev) * 0.02).to(a_dt) if self.with_scaling else None
        zeros = None
        if self.with_zeros:
            zeros = (torch.zeros(self.K // g, self.N * self.bit // 8, device=dev, dtype=torch.int8) if self.zeros_mode == "quantized"
                     else torch.full((self.N, self.K // g), float(1 << (self.bit - 1)), device=dev).to(a_dt))
        lut = self._ensure_lut(dev)
        stream = _lib.current_stream_handle(dev)

        def time_at(m, min_m):
            self._desc.two_pass_min_m = min_m
            plan = self.lib.plan(m)                          # re-plans (and drops the cached scratch size)
            if min_m and plan["kernel_family"] != 4:
                return None
            A = (torch.rand(m, self._a_cols, device=dev) - 0.5).to(a_dt) if a_dt.is_floating_point else \
                torch.randint(-128, 127, (m, self._a_cols), device=dev, dtype=a_dt)
            out = torch.empty((m, self.N), dtype=self.torch_output_dtype, device=dev)
            args = (A.data_ptr(), W.data_ptr(), lut.data_ptr() if lut is not None else None,
                    scale.data_ptr() if scale is not None else None, zeros.data_ptr() if zeros is not None else None, None,
                    out.data_ptr(), m, stream, dev)
            for _ in range(2):
                self.lib.run(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                self.lib.run(*args)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / iters

        threshold = 0
        for m in reversed(cand):                             # the two-pass member's advantage grows with M
            fused, two = time_at(m, 0), time_at(m, 1)
            if two is None or two > 0.97 * fused:
                break
            threshold = m
            self._tuned = getattr(self, "_tuned", {})
            self._tuned[m] = {"fused_ms": fused, "two_pass_ms": two}
        self._desc.two_pass_min_m = threshold
        self.lib._ws.clear()                                 # scratch buffers of the losing member
        self.lib.plan(cand[-1])

    def get_source(self, *args, **kwargs) -> str:
        names = sorted({p["name"] for p in self.plans.values()})
        return "// prebuilt gfx950 kernels: " + ", ".join(names)

    @property
    def libpath(self):
        return _lib.LIB_PATH

    @property
    def srcpath(self):
        import os
        return os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")

    def update_runtime_module(self, rt_mod=None, srcpath=None, libpath=None):
        """Nothing to swap in: the kernels are prebuilt (ops/operator.py:471-485)."""
        self.lib.init()

    def retrieve_weight_shape(self):
        """Shape of the B operand the kernels expect = `transform_weight` output (:645-660)."""
        if self.bit < 8 and self.bit in (1, 2, 4):
            return [self.N, self.K * self.bit // 8]
        return [self.N, self.K]

    def _ensure_lut(self, device):
        if self.source_format != "nf":
            return None
        if self.lut is None or self._lut_device != device:
            self.lut = torch.tensor(self.NF4_VALUES, dtype=torch_dtype(self.A_dtype), device=device)
            self._lut_device = device
            self.lib.default_lut = self.lut
        return self.lut

    def transform_weight(self, weight, scale=None, zeros=None, bias=None):
        """Quantised integer / fp8 weight (N, K) -> kernel operand, on the CPU (:662-711).

        int formats below 8 bit: clamp to [-2^(b-1), 2^(b-1)], shift to unsigned codes, bit-pack
        (little field first), then LOP3-interleave when `fast_decoding`.  The bytes are exactly the
        reference's checkpoint layout.
        """
        weight = weight.contiguous()
        if self.W_dtype == self.A_dtype:
            if self.weight_transform is not None:
                return self.weight_transform(weight.cpu()).to(weight.device).contiguous()
            return weight
        if self.source_format == "int" and self.bit < 8:
            assert not self.with_scaling, "scale should be False for int source format"
            assert not self.with_zeros, "zeros should be False for int source format"
            maxq = 2 ** (self.bit - 1)
            if weight.dtype == torch.int8 and not weight.is_cuda:
                # the same three ops through numpy: torch's int8 clamp / add on the CPU run at ~6 ns per element (100 ms for a
                # 4096 x 4096 layer), numpy's are vectorised (3 ms)
                import numpy as np
                weight = torch.from_numpy(np.clip(weight.numpy(), -maxq, maxq) + np.int8(maxq))
            else:
                weight = torch.clamp(weight, -maxq, maxq).char() + maxq
        elif self.source_format in ("fp_e5m2", "fp_e4m3"):
            weight = weight.view(torch.int8)
        else:
            weight = weight.char()
        if self.weight_transform is not None:
            weight = self.weight_transform(weight.cpu()).to(weight.device).contiguous()
        result = [weight]
        for extra in (scale, zeros, bias):
            if extra is not None:
                result.append(extra)
        return next(iter(result), result)

    def dequantize_weight(self, W, scale=None, zeros=None, out=None):
        """`B_decode` of the TE graph (tirscript/matmul_dequantize_impl.py:391-449) as a tensor: `(N, K)` in A_dtype, every
        weight decoded and (zero, scale)-dequantised by the routines the MFMA members use in their loop (`wqaa_dequantize`;
        bit-identical to the oracle's `dequantize_weight`, tests/test_two_pass_gpu.py).  float16 / bfloat16 / int8
        activations' operators, K a multiple of 128 (256 for int8).  Asynchronous on the current stream of W's device."""
        import ctypes
        if not W.is_cuda:
            raise RuntimeError("bitblas_amd.Matmul runs on the GPU only (no CPU fallback)")
        if W.numel() * W.element_size() != self._w_bytes:
            raise ValueError(f"W holds {W.numel() * W.element_size()} bytes, the operator expects {self._w_bytes}")
        if out is None:
            out = torch.empty((self.N, self.K), dtype=self._a_torch_dtype, device=W.device)
        elif tuple(out.shape) != (self.N, self.K) or out.dtype != self._a_torch_dtype or not out.is_contiguous() or out.device != W.device:
            raise ValueError(f"out must be a contiguous ({self.N}, {self.K}) {self._a_torch_dtype} tensor on W's device")
        lut = self._ensure_lut(W.device)
        L = _lib.load_library()
        _lib.check(L.wqaa_dequantize(ctypes.byref(self._desc), W.data_ptr(), lut.data_ptr() if lut is not None else None,
                                     scale.data_ptr() if scale is not None else None,
                                     zeros.data_ptr() if zeros is not None else None,
                                     out.data_ptr(), _lib.current_stream_handle(W.device)))
        return out

    def transform_input(self, input_tensor):
        return input_tensor  # propagate_a is always NonTransform on CDNA (see propagate_a)

    def check_activation(self, A) -> int:
        """What the kernels assume about A and the reference never checks (it passes `data_ptr()` on,
        ops/operator.py:458-463): device memory, K columns, the operator's A_dtype, and for a static-M operator
        exactly M rows - a short A would be read out of bounds.  Returns the row count m."""
        if not A.is_cuda:
            raise RuntimeError("bitblas_amd.Matmul runs on the GPU only (no CPU fallback)")
        if A.shape[-1] != self._a_cols:
            raise ValueError(f"A has {A.shape[-1]} columns, the operator was built for {self._a_cols} (K={self.K})")
        if A.dtype != self._a_torch_dtype:
            raise TypeError(f"A is {A.dtype}, the operator was built for A_dtype={self.A_dtype} ({self._a_torch_dtype})")
        m = A.numel() // self._a_cols
        if self.dynamic_range is None and m != self.config.M:
            raise ValueError(f"operator was built for M={self.config.M}, got {m} rows")
        return m

    def check_output(self, output, m: int):
        """a caller's `output` receives m * N elements of out_dtype through a raw pointer (the reference passes `data_ptr()`
        on unchecked, ops/operator.py:458-463): a smaller or differently-typed buffer is an out-of-bounds write"""
        if output.dtype != self.torch_output_dtype or output.numel() != m * self.N:
            raise ValueError(f"output must hold {m} x {self.N} {self.torch_output_dtype} elements, got {tuple(output.shape)} {output.dtype}")

    def forward(self, A, W, scale=None, zeros=None, bias=None, output=None) -> Any:
        """`matmul(A, W, scale, zeros, bias, output)` (:724-753).  Launches on the current stream
        of A's device; returns immediately (asynchronous)."""
        m = self.check_activation(A)
        if output is None:
            output = torch.empty(A.shape[:-1] + (self.N,), dtype=self.torch_output_dtype, device=A.device)
        elif not output.is_contiguous():
            raise ValueError("output must be a contiguous tensor")
        else:
            self.check_output(output, m)
        if W.numel() * W.element_size() != self._w_bytes:
            raise ValueError(f"W holds {W.numel() * W.element_size()} bytes, the operator expects {self._w_bytes} "
                             f"(shape {self.retrieve_weight_shape()}: run transform_weight first)")
        if not A.is_contiguous():
            A = A.contiguous()   # the kernels read raw row-major memory (upstream passes data_ptr() unchecked)
        lut = self._ensure_lut(A.device)
        stream = _lib.current_stream_handle(A.device)
        self.lib.run(
            A.data_ptr(), W.data_ptr(), lut.data_ptr() if lut is not None else None,
            scale.data_ptr() if scale is not None else None,
            zeros.data_ptr() if zeros is not None else None,
            bias.data_ptr() if bias is not None else None,
            output.data_ptr(), m, stream, A.device)
        return output

    __call__ = forward

    def fused_ops_supported(self, m: int) -> bool:
        """whether the caller's elementwise ops fold into this operator's launch at this row count (`forward_ex`,
        `bitblas_amd.matmul_gate_up`): float16 activations x 1 / 2 / 4-bit integer weights, float16 output, m <= 2
        (include/wqaa.h, WQAA_EPI_ADD_RESIDUAL / wqaa_matmul_gate_up)"""
        cfg = self.config
        elems = 128 // self.bit if self.bit in (1, 2, 4) else 0
        group = cfg.K if (cfg.group_size or -1) <= 0 else cfg.group_size
        return (1 <= m <= 2 and cfg.A_dtype == "float16" and cfg.out_dtype == "float16" and self.source_format in ("int", "uint")
                and elems > 0 and cfg.K % elems == 0 and cfg.K % group == 0 and (not cfg.with_scaling or group % elems == 0))

    def norm_supported(self, m: int) -> bool:
        """whether the RMSNorm in front of this operator folds into its launch at this row count (`forward_ex(norm=...)`,
        `matmul_group(norm=...)`, `matmul_gate_up(norm=...)`): `fused_ops_supported` and the activation rows within the registers
        a workgroup loads ahead (include/wqaa.h WQAA_EPI_RMSNORM_INPUT).  A quick necessary condition (K <= 12288 at 4 bit: three
        items per thread of a 16-wave workgroup); the selector has the last word and the entries fall back on its refusal."""
        return self.fused_ops_supported(m) and m * self.config.K * self.bit <= 12288 * 4

    def forward_ex(self, A, W, scale=None, zeros=None, bias=None, output=None, residual=None, norm=None) -> Any:
        """`forward` with one of the elementwise ops a decoder layer runs next to it (the reference's callers:
        integration/BitNet/modeling_bitnet.py:839-860): `residual` - `residual + forward(A, ...)` (may be `output` itself);
        `norm` = (weight, eps) - `forward(rms_norm(A), ...)`, A the hidden state in front of the layer's RMSNorm (:89-104).
        One launch where `fused_ops_supported(m)` / `norm_supported(m)`; elsewhere torch's kernels around `forward`."""
        if residual is None and norm is None:
            return self.forward(A, W, scale, zeros, bias, output)
        if residual is not None and norm is not None:
            raise ValueError("residual add and RMSNorm input on one projection: no layer has both")
        m = self.check_activation(A)
        if residual is not None and (residual.dtype != self.torch_output_dtype or residual.numel() != m * self.N or residual.device != A.device):
            raise ValueError(f"`residual` must hold {m} x {self.N} {self.torch_output_dtype} elements on A's device")
        if norm is not None:
            check_norm(norm, A, self.K)
        fused = self.norm_supported(m) if norm is not None else self.fused_ops_supported(m)
        if not fused:
            if norm is not None:
                return self.forward(rms_norm_reference(A, *norm), W, scale, zeros, bias, output)
            if output is not None and output.data_ptr() == residual.data_ptr():
                residual = residual.clone()
            out = self.forward(A, W, scale, zeros, bias, output)
            out += residual.view_as(out)
            return out
        if output is None:
            output = torch.empty(A.shape[:-1] + (self.N,), dtype=self.torch_output_dtype, device=A.device)
        elif not output.is_contiguous() or output.device != A.device:
            raise ValueError("output must be a contiguous tensor on A's device")      # (handed to the kernel as a raw pointer)
        else:
            self.check_output(output, m)
        if W.numel() * W.element_size() != self._w_bytes:
            raise ValueError(f"W holds {W.numel() * W.element_size()} bytes, the operator expects {self._w_bytes} "
                             f"(shape {self.retrieve_weight_shape()}: run transform_weight first)")
        A = A if A.is_contiguous() else A.contiguous()
        residual = residual if residual is None or residual.is_contiguous() else residual.contiguous()
        try:
            self.lib.run_residual(
                A.data_ptr(), W.data_ptr(), scale.data_ptr() if scale is not None else None,
                zeros.data_ptr() if zeros is not None else None, bias.data_ptr() if bias is not None else None,
                output.data_ptr(), m, _lib.current_stream_handle(A.device),
                residual=residual.data_ptr() if residual is not None else None,
                norm=(norm[0].data_ptr(), norm[1]) if norm is not None else None)
        except _lib.WqaaError as exc:
            # the selector's word is final (e.g. two activation rows of K = 8192 do not fit the registers a narrower workgroup loads
            # ahead): nothing was launched - the reference's norm as torch kernels in front of the plain launch
            if norm is None or exc.code != _lib.ERR_UNSUPPORTED:
                raise
            return self.forward(rms_norm_reference(A, *norm), W, scale, zeros, bias, output)
        return output

    def _forward_from_prebuild_lib(self, *args, stream=0):
        """ops/operator.py:458-463"""
        self.lib.call(*args, stream)

    def profile_latency(self, dynamic_symbolic_constraints=None, iters: int = 50) -> float:
        """Mean kernel latency in ms on synthetic operands (reference: time_evaluator, :442-450)."""
        m = self.config.M if isinstance(self.config.M, int) else (
            (dynamic_symbolic_constraints or {}).get("m", self.config.M[0]))
        dev = self.device or torch.device("cuda")
        a_dt = torch_dtype(self.A_dtype)
        k_cols = self.K // 2 if self.A_dtype == "int4" else self.K      # int4 activations: two per byte
        A = (torch.rand(m, k_cols, device=dev) - 0.5).to(a_dt) if a_dt.is_floating_point else \
            torch.randint(-8, 8, (m, k_cols), device=dev, dtype=a_dt)
        W = torch.randint(-128, 127, self.retrieve_weight_shape(), device=dev, dtype=torch.int8)
        if self.W_dtype == self.A_dtype and self.bit >= 8:
            W = A.new_zeros((self.N, self.K))      # dense pair; sub-byte native pairs (int4 x int4) keep the packed shape
        g = self.K if self.group_size in (-1, None) else self.group_size
        scale = torch.rand(self.N, self.K // g, device=dev).to(a_dt) if self.with_scaling else None
        zeros = None
        if self.with_zeros:
            if self.zeros_mode == "quantized":
                zeros = torch.zeros(self.K // g, self.N * self.bit // 8, device=dev, dtype=torch.int8)
            else:
                zeros = torch.zeros(self.N, self.K // g, device=dev).to(a_dt)
        bias = torch.zeros(self.N, device=dev).to(a_dt) if self.with_bias else None
        for _ in range(5):
            self.forward(A, W, scale, zeros, bias)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            self.forward(A, W, scale, zeros, bias)
        stop.record()
        stop.synchronize()
        return start.elapsed_time(stop) / iters

    def cleanup(self):
        self.workspace = None

    # -- config passthroughs ---------------------------------------------------------------------
    M = property(lambda self: self.config.M)
    N = property(lambda self: self.config.N)
    K = property(lambda self: self.config.K)
    A_dtype = property(lambda self: self.config.A_dtype)
    W_dtype = property(lambda self: self.config.W_dtype)
    out_dtype = property(lambda self: self.config.out_dtype)
    accum_dtype = property(lambda self: self.config.accum_dtype)
    storage_dtype = property(lambda self: self.config.storage_dtype)
    with_scaling = property(lambda self: self.config.with_scaling)
    with_zeros = property(lambda self: self.config.with_zeros)
    group_size = property(lambda self: self.config.group_size)
    fast_decoding = property(lambda self: self.config.fast_decoding)
    with_bias = property(lambda self: self.config.with_bias)
    layout = property(lambda self: self.config.layout)
    zeros_mode = property(lambda self: self.config.zeros_mode)

    @property
    def propagate_a(self):
        # the reference returns NonTransform whenever the arch has no NVIDIA mma (:808-820);
        # that is every CDNA part, so no ladder layout ever reaches our kernels
        return TransformKind.NonTransform

    @property
    def propagate_b(self):
        return TransformKind.NonTransform

    @property
    def input_transform(self):
        return self.input_executors if self.input_executors.size else None

    @property
    def weight_transform(self):
        return self.weight_executors if self.weight_executors.size else None


@dataclass(frozen=True)
class MatmulConfigWithSplitK(MatmulConfig):
    """`bitblas.MatmulConfigWithSplitK` (ops/general_matmul_splitk.py:21-23)."""
    k_split: int = 1  # split K dimension


class MatmulWithSplitK(Matmul):
    """`bitblas.MatmulWithSplitK` (ops/general_matmul_splitk.py:26-199).

    Upstream launches a kernel that writes `k_split` partial products in out_dtype and reduces them with
    `torch.sum` (:168-186).  Here `k_split` travels to the tile selector as `wqaa_matmul_desc.k_split_hint`: it sets
    the K split across the waves of a workgroup of the M <= 2 exact-product GEMV (`strict_reference=False`) and the
    split-K count of the pipelined MFMA members; the one-launch decode member (M <= 16) and the skinny member
    (M <= 64) split K structurally and keep their own count - `plans[m]["split_k"]` says what was taken.  Partial
    sums are fp32 / int32 with one rounding at the end, so the result is at least as accurate as upstream's sum of
    rounded partials.  Weight layout, arguments and return value are those of `Matmul`."""

    @property
    def k_split(self):
        return getattr(self.config, "k_split", 1)
