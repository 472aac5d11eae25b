"""Test helpers with the reference's semantics (bitblas/testing/__init__.py:29-91)."""
from __future__ import annotations

import inspect
import sys


def main():
    """Run the calling test file through pytest (reference: testing/__init__.py:13-15)."""
    import pytest
    test_file = inspect.getsourcefile(sys._getframe(1))
    sys.exit(pytest.main([test_file] + sys.argv[1:]))


def torch_assert_close(tensor_a, tensor_b, rtol=1e-2, atol=1e-3, max_mismatched_ratio=0.001,
                       verbose=False):
    """Pass when at most `max_mismatched_ratio` of the elements violate
    |a - b| <= atol + rtol * |b|; raises AssertionError otherwise."""
    import torch
    a = tensor_a.to(torch.float64)
    b = tensor_b.to(torch.float64).to(a.device)
    diff = (a - b).abs()
    bad = int((diff > atol + rtol * b.abs()).sum().item())
    total = a.numel()
    allowed = int(total * max_mismatched_ratio)
    if verbose:
        print(f"Number of mismatched elements: {bad} / {total} (allowed: {allowed})")
    if bad > allowed:
        raise AssertionError(
            f"Too many mismatched elements: {bad} > {allowed} "
            f"({max_mismatched_ratio * 100:.2f}% allowed, but get {bad / total * 100:.2f}%). "
            f"Greatest absolute difference: {diff.max().item()}, "
            f"Greatest relative difference: {(diff / (b.abs() + 1e-12)).max().item()}.")
    return True
