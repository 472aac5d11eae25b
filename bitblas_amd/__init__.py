"""bitblas_amd - an MI355X (gfx950) native backend behind the microsoft/BitBLAS operator API.

    import bitblas_amd as bitblas
    matmul = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=4096, K=4096, A_dtype="float16",
                                                 W_dtype="int4", ...))

Public names follow `bitblas/__init__.py:155-175`.  Everything below the operator API is different:
hand-written HIP kernels in one prebuilt shared library (`libwqaa_hip.so`, C ABI in
`include/wqaa.h`) instead of TVM/TileLang code generation.
"""
from __future__ import annotations

import logging

__version__ = "0.1.0"

from .target import auto_detect_nvidia_target, auto_detect_target, get_arch  # noqa: F401
from .matmul import (  # noqa: F401
    Matmul, MatmulConfig, MatmulConfigWithSplitK, MatmulKernelNameGenerator, MatmulWithSplitK, Operator, OperatorConfig,
    OptimizeStrategy, TransformKind, is_native_compute,
)
from .module import Linear  # noqa: F401
from .group import GatedMLP, LinearGroup, gate_up_plan, group_plan, matmul_gate_up, matmul_group  # noqa: F401  (MI355X extension: one launch for q/k/v, gate/up)
from .chain import ChainStep, DecoderTail, chain_plan, matmul_chain  # noqa: F401  (a chain of dependent operators described once, run as its launches)
from .cache import (  # noqa: F401
    OperatorCache, get_database_path, global_operator_cache, load_global_ops_cache,
    set_database_path,
)
from . import quantization, testing  # noqa: F401
from .quantization import general_compress, interleave_weight  # noqa: F401
from .compat import install_as_bitblas  # noqa: F401  (`import bitblas` and its submodule paths answered by this package)


def set_log_level(level):
    """`bitblas.set_log_level` (bitblas/__init__.py:28-55): accepts a name or a logging level."""
    if isinstance(level, str):
        level = getattr(logging, level.upper(), logging.INFO)
    logging.getLogger(__name__).setLevel(level)


logging.getLogger(__name__).addHandler(logging.NullHandler())
set_log_level("WARNING")
