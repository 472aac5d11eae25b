"""Generate golden `MatmulConfig` legalisation vectors by RUNNING the reference's own class.

`import bitblas` is impossible here (tvm / tilelang submodules are empty), but `MatmulConfig` and
`MatmulKernelNameGenerator` (bitblas/ops/general_matmul/__init__.py:58-318) are plain Python: this
script slices their source text out of the reference checkout AT RUN TIME (nothing is copied into
the repo), executes it in a namespace holding the reference's own `TransformKind` /
`OptimizeStrategy` (bitblas/base/operator_common.py, loaded standalone), and records, for a grid of
constructor arguments, `repr(config)` (the operator-cache key, cache/operator.py:62), every field
after `__post_init__`, and the default kernel name.

Output: tests/golden/matmul_config_golden.json.  Test infrastructure only.
"""
from __future__ import annotations

import dataclasses
import importlib.util
import itertools
import json
import logging
import os
import re
import sys
import typing

REF = "/root/reference/bitblas"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "matmul_config_golden.json")


def load_reference_classes():
    spec = importlib.util.spec_from_file_location("ref_operator_common", os.path.join(REF, "base", "operator_common.py"))
    common = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(common)
    src = open(os.path.join(REF, "ops", "general_matmul", "__init__.py")).read()
    start = src.index("@dataclass(frozen=True)\nclass MatmulConfig")
    end = src.index("class Matmul(Operator):")
    body = src[start:end]
    op_src = open(os.path.join(REF, "ops", "operator.py")).read()
    m = re.search(r"class BaseKernelNameGenerator.*?(?=\nclass |\Z)", op_src, flags=re.S)

    @dataclasses.dataclass(frozen=True)
    class OperatorConfig:
        pass

    class Hint:  # never instantiated: generate(hint=None) only
        pass

    ns = {"dataclass": dataclasses.dataclass, "OperatorConfig": OperatorConfig, "Hint": Hint,
          "TransformKind": common.TransformKind, "OptimizeStrategy": common.OptimizeStrategy,
          "logger": logging.getLogger("ref"), "CONFIG_INFO_MESSAGE_STRATEGY": "{}",
          "re": re, "ABC": __import__("abc").ABC, "abstractmethod": __import__("abc").abstractmethod}
    ns.update({k: getattr(typing, k) for k in ("Any", "Literal", "Optional", "Tuple", "Union", "List", "Dict")})
    exec(m.group(0), ns)
    exec(body, ns)
    return ns["MatmulConfig"], ns["MatmulKernelNameGenerator"], common


def grid():
    base = dict(N=1024, K=1024)
    cases = []
    for M in (1, 16, 768, [1, 16, 64], None):
        for A, W in (("float16", "float16"), ("float16", "int4"), ("float16", "uint4"), ("float16", "int2"),
                     ("float16", "nf4"), ("float16", "fp4_e2m1"), ("float16", "e4m3_float8"), ("int8", "int8"),
                     ("int8", "int2"), ("int8", "int4"), ("bfloat16", "uint4"), ("e4m3_float8", "e4m3_float8"),
                     ("float16", "int8"), ("float16", "uint1")):
            cases.append(dict(base, M=M, A_dtype=A, W_dtype=W))
    for extra in (dict(fast_decoding=False), dict(fast_decoding=True), dict(propagate_a=False, propagate_b=False),
                  dict(propagate_b=True), dict(propagate_a=True, propagate_b=True), dict(optimize_stratety=1),
                  dict(with_scaling=True, with_zeros=True, zeros_mode="quantized", group_size=128),
                  dict(with_bias=True, group_size=32, with_scaling=True), dict(zeros_mode=None, fast_decoding=None),
                  dict(N=1000, K=1024), dict(N=1024, K=1000), dict(accum_dtype="float32", out_dtype="float32")):
        for M in (1, 256, [16, 32]):
            for W in ("uint4", "float16", "int2"):
                kw = dict(base, M=M, A_dtype="float16", W_dtype=W)
                kw.update(extra)
                cases.append(kw)
    return cases


def main():
    if not os.path.isdir(REF):
        print("reference not present; golden vectors are already committed", file=sys.stderr)
        return 0
    MatmulConfig, NameGen, _ = load_reference_classes()
    out = []
    for kw in grid():
        cfg = MatmulConfig(**kw)
        fields = {}
        for f in dataclasses.fields(cfg):
            v = getattr(cfg, f.name)
            fields[f.name] = int(v) if hasattr(v, "value") and not isinstance(v, (bool, int)) or type(v).__name__ in ("TransformKind", "OptimizeStrategy") else (list(v) if isinstance(v, tuple) else v)
        out.append({"kwargs": kw, "repr": repr(cfg), "fields": fields, "kernel_name": NameGen(cfg).generate()})
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
    print(f"wrote {OUT}: {len(out)} configs")
    return 0


if __name__ == "__main__":
    sys.exit(main())
