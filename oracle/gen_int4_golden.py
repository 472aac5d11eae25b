"""Golden vectors for packed int4 activations, produced by RUNNING the reference's own int4 operator test.

`testing/python/operators/test_general_matmul_ops_int4.py::matmul_int4_torch_forward` builds int4 x int4 and
int4 x int2 operands, packs them by hand (:44-57) and compares the operator with `A.float() @ B.T.float()`
(:141-146).  The function is executed from the file where it lies, with a recorder standing in for `bitblas`
(see gen_optest_golden.py), for the two argument tuples of its test list that need no TVM operator
(:150 int4 x int4 and :152 int4 x int2, both propagate_b=False, fast_decoding=False; the other four build
LadderPermutate / LOP3Permutate TVM ops).  The test is unseeded upstream: `torch.manual_seed(0)` is set before
each call; `device="cuda"` keyword arguments of torch.randint / torch.zeros are dropped (no GPU here).

Output: tests/golden/int4_golden.npz: packed A, packed B exactly as the test hands them to the operator, and the
test's expected int32 result.  Runs only where /root/reference exists.  Test infrastructure only.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REF_FILE = "/root/reference/testing/python/operators/test_general_matmul_ops_int4.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "int4_golden.npz")


def main():
    if not os.path.exists(REF_FILE):
        print("reference not present; golden vectors are already committed", file=sys.stderr)
        return 0
    import torch
    cases, cur = [], {}

    class Matmul:
        def __init__(self, config=None, enable_tuning=False, **_):
            cur.clear()
            cur["config"] = config

        def get_source(self):
            return ""

        def profile_latency(self):
            return 0.0

        def __call__(self, a, b, output=None):
            cur["A"], cur["B"], cur["out"] = a, b, output

    def record(actual, expected, **_):
        assert actual is cur["out"]
        cases.append(dict(config=cur["config"], A=cur["A"], B=cur["B"], expected=expected))

    bb = types.ModuleType("bitblas")
    bb.__path__ = []
    bb.Matmul = Matmul
    bb.MatmulConfig = lambda **kw: dict(kw)
    bb.set_log_level = lambda *a, **k: None
    testing = types.ModuleType("bitblas.testing")
    testing.main = lambda: None
    bb.testing = testing
    sys.modules.update({"bitblas": bb, "bitblas.testing": testing})
    torch.testing.assert_close = record
    for fn_name in ("randint", "zeros"):
        real = getattr(torch, fn_name)
        setattr(torch, fn_name, (lambda real: lambda *a, **k: real(*a, **{x: y for x, y in k.items() if x != "device"}))(real))
    ns = {"__name__": "ref_int4_optest", "__file__": REF_FILE, "print": lambda *a, **k: None}
    exec(compile(open(REF_FILE).read(), REF_FILE, "exec"), ns)
    fn = ns["matmul_int4_torch_forward"]
    for args in ((128, 128, 128, "int4", "int4", "int32", "int32", "nt", False),          # :150
                 (128, 128, 128, "int4", "int2", "int32", "int32", "nt", False, False)):  # :152
        torch.manual_seed(0)
        fn(*args)
    out = {}
    for i, c in enumerate(cases):
        out[f"c{i}_W_dtype"] = np.array(c["config"]["W_dtype"])
        out[f"c{i}_A"] = c["A"].numpy()
        out[f"c{i}_B"] = c["B"].numpy()
        out[f"c{i}_expected"] = c["expected"].numpy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {len(cases)} cases to {OUT}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
