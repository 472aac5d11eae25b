"""`import bitblas` answered by this package: the import paths the reference's callers use.

A user of microsoft/BitBLAS does not only write `import bitblas; bitblas.Matmul(...)`.  The callers either side of the
hot path import from the reference's SUBMODULES (SURVEY.md section 8b "Callers", section 8f):

    integration/BitNet/utils_quant.py:9-11        from bitblas.cache import global_operator_cache, get_database_path
                                                  from bitblas import Matmul, MatmulConfig, auto_detect_nvidia_target
    integration/pytorch/bitblas_quant_linear.py   from bitblas.quantization.utils import general_compress, interleave_weight
                                                  from bitblas.utils import auto_detect_nvidia_target
    integration/BitNet/vllm_workspace, AutoGPTQ / GPTQModel / vLLM quant layers (README pointers): the same names plus
                                                  from bitblas.ops import Operator, Matmul, MatmulConfig
                                                  from bitblas.module import Linear, unpack_qweight, unpack_qzeros
    testing/python/**                             import bitblas.testing; bitblas.testing.torch_assert_close / main

`install_as_bitblas()` registers module objects under those dotted names (`bitblas`, `bitblas.ops`,
`bitblas.ops.general_matmul`, `bitblas.ops.general_matmul_splitk`, `bitblas.ops.operator`, `bitblas.ops.common`,
`bitblas.cache`, `bitblas.cache.operator`, `bitblas.module`, `bitblas.quantization`, `bitblas.quantization.utils`,
`bitblas.utils`, `bitblas.utils.target_detector`, `bitblas.testing`, `bitblas.common`) whose attributes are THIS
package's objects, so a caller written against the reference runs unmodified on MI355X.  What the reference exports but
only its code generator needs (`tvm`, `tilelang`, `base`, `gpu`, `relax`, `tl`, `builder`) is deliberately absent:
importing it fails with an ImportError that says so, instead of pretending.

Nothing here touches the hot path: it is `sys.modules` bookkeeping, done once.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types
from typing import Dict, Optional

# code-generation subpackages of the reference that have no counterpart here (OUT OF SCOPE)
_CODEGEN_ONLY = ("tvm", "tilelang", "base", "gpu", "relax", "tl", "builder", "benchmark")

_installed: Dict[str, Dict[str, Optional[types.ModuleType]]] = {}


def _module(name: str, doc: str, package: bool = False, **attrs) -> types.ModuleType:
    m = types.ModuleType(name, doc)
    if package:
        m.__path__ = []          # a package: `import bitblas.ops.general_matmul` resolves through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    m.__all__ = sorted(attrs)
    return m


class _RefusingFinder(importlib.abc.MetaPathFinder):
    """`import bitblas.tvm` & co: a clear ImportError instead of `No module named ...` (or, worse, a half-working stub)"""

    def __init__(self, alias: str):
        self.alias = alias

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(self.alias + "."):
            return None
        head = fullname[len(self.alias) + 1:].split(".")[0]
        if head in _CODEGEN_ONLY:
            raise ImportError(
                f"{fullname}: the MI355X backend (bitblas_amd) has no code generator - kernels are prebuilt HIP "
                f"(bitblas_amd/libwqaa_hip.so, include/wqaa.h); `{self.alias}.{head}` exists only in microsoft/BitBLAS")
        return None


def alias_modules(alias: str = "bitblas") -> Dict[str, types.ModuleType]:
    """the dotted-name -> module map `install_as_bitblas` registers (built fresh; nothing is registered here)"""
    import bitblas_amd as pkg
    from . import cache, matmul, module, quantization, target, testing

    top = _module(alias, pkg.__doc__ or "", package=True)
    for k in dir(pkg):
        if not k.startswith("__"):
            setattr(top, k, getattr(pkg, k))
    top.__version__ = pkg.__version__
    top.__backend__ = "bitblas_amd"

    general_matmul = _module(
        f"{alias}.ops.general_matmul", "bitblas/ops/general_matmul/__init__.py", package=True,
        Matmul=matmul.Matmul, MatmulConfig=matmul.MatmulConfig, MatmulKernelNameGenerator=matmul.MatmulKernelNameGenerator,
        is_native_compute=matmul.is_native_compute, OptimizeStrategy=matmul.OptimizeStrategy,
        TransformKind=matmul.TransformKind, OperatorConfig=matmul.OperatorConfig, Operator=matmul.Operator,
        OPExecutorCPU=matmul.OPExecutorCPU)
    splitk = _module(
        f"{alias}.ops.general_matmul_splitk", "bitblas/ops/general_matmul_splitk.py",
        MatmulConfigWithSplitK=matmul.MatmulConfigWithSplitK, MatmulWithSplitK=matmul.MatmulWithSplitK,
        MatmulConfig=matmul.MatmulConfig, Matmul=matmul.Matmul)
    operator = _module(
        f"{alias}.ops.operator", "bitblas/ops/operator.py",
        Operator=matmul.Operator, OperatorConfig=matmul.OperatorConfig, OPExecutorCPU=matmul.OPExecutorCPU,
        BaseKernelNameGenerator=matmul.BaseKernelNameGenerator, TransformKind=matmul.TransformKind)
    common = _module(
        f"{alias}.ops.common", "bitblas/ops/common.py",
        OptimizeStrategy=matmul.OptimizeStrategy, TransformKind=matmul.TransformKind)
    ops = _module(
        f"{alias}.ops", "bitblas/ops/__init__.py", package=True,
        Operator=matmul.Operator, OperatorConfig=matmul.OperatorConfig, Matmul=matmul.Matmul, MatmulConfig=matmul.MatmulConfig,
        general_matmul=general_matmul, general_matmul_splitk=splitk, operator=operator, common=common)

    cache_operator = _module(
        f"{alias}.cache.operator", "bitblas/cache/operator.py",
        OperatorCache=cache.OperatorCache, global_operator_cache=cache.global_operator_cache,
        load_global_ops_cache=cache.load_global_ops_cache, get_database_path=cache.get_database_path,
        set_database_path=cache.set_database_path)
    cache_pkg = _module(
        f"{alias}.cache", "bitblas/cache/__init__.py", package=True,
        OperatorCache=cache.OperatorCache, global_operator_cache=cache.global_operator_cache,
        load_global_ops_cache=cache.load_global_ops_cache, get_database_path=cache.get_database_path,
        set_database_path=cache.set_database_path, operator=cache_operator)

    module_pkg = _module(
        f"{alias}.module", "bitblas/module/__init__.py", package=True,
        Linear=module.Linear, unpack_qzeros=module.unpack_qzeros, unpack_qzeros_v2=module.unpack_qzeros_v2,
        unpack_qweight=module.unpack_qweight)

    q_utils = _module(
        f"{alias}.quantization.utils", "bitblas/quantization/utils.py",
        general_compress=quantization.general_compress, interleave_weight=quantization.interleave_weight,
        gen_quant4=quantization.gen_quant4)
    q_pkg = _module(
        f"{alias}.quantization", "bitblas/quantization/__init__.py", package=True,
        general_compress=quantization.general_compress, interleave_weight=quantization.interleave_weight,
        gen_quant4=quantization.gen_quant4, utils=q_utils)

    detector = _module(
        f"{alias}.utils.target_detector", "bitblas/utils/target_detector.py",
        auto_detect_nvidia_target=target.auto_detect_nvidia_target, auto_detect_target=target.auto_detect_target)
    utils = _module(
        f"{alias}.utils", "bitblas/utils/__init__.py", package=True,
        auto_detect_nvidia_target=target.auto_detect_nvidia_target, auto_detect_target=target.auto_detect_target,
        get_default_cache_path=cache.get_database_path, target_detector=detector)

    common_top = _module(
        f"{alias}.common", "bitblas/common.py",
        BITBLAS_DEFAULT_CACHE_PATH=cache.BITBLAS_DEFAULT_CACHE_PATH, MAX_ERROR_MESSAGE_LENGTH=500)

    testing_mod = _module(
        f"{alias}.testing", "bitblas/testing/__init__.py", package=True,
        main=testing.main, torch_assert_close=testing.torch_assert_close)

    top.ops, top.cache, top.module, top.quantization = ops, cache_pkg, module_pkg, q_pkg
    top.utils, top.testing, top.common = utils, testing_mod, common_top
    mods = {m.__name__: m for m in (top, ops, general_matmul, splitk, operator, common, cache_pkg, cache_operator, module_pkg,
                                    q_pkg, q_utils, utils, detector, common_top, testing_mod)}
    return mods


def install_as_bitblas(alias: str = "bitblas", force: bool = False) -> types.ModuleType:
    """Make `import <alias>` (default `bitblas`) and its submodule imports resolve to this package.

    Refuses when a DIFFERENT module already answers to the name (a real BitBLAS installation that was imported first)
    unless `force=True`; calling it twice is a no-op.  Returns the top-level alias module."""
    if alias in _installed:
        return sys.modules[alias]
    present = sys.modules.get(alias)
    if present is not None and getattr(present, "__backend__", None) != "bitblas_amd" and not force:
        raise RuntimeError(f"a module named {alias!r} is already imported ({getattr(present, '__file__', present)}); "
                           f"pass force=True to replace it for this process")
    mods = alias_modules(alias)
    saved = {}
    for name, m in mods.items():
        saved[name] = sys.modules.get(name)
        sys.modules[name] = m
    finder = _RefusingFinder(alias)
    sys.meta_path.insert(0, finder)
    saved["__finder__"] = finder
    _installed[alias] = saved
    return mods[alias]


def uninstall(alias: str = "bitblas") -> None:
    """undo `install_as_bitblas` (tests)"""
    saved = _installed.pop(alias, None)
    if saved is None:
        return
    finder = saved.pop("__finder__", None)
    if finder in sys.meta_path:
        sys.meta_path.remove(finder)
    for name, old in saved.items():
        if old is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = old
