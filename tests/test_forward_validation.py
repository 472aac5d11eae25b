"""What `Matmul.forward` / `Linear.forward` check before handing raw pointers to a kernel: the reference
passes `data_ptr()` on unchecked (bitblas/ops/operator.py:458-463, bitblas/module/__init__.py:267-289); a wrong shape
there is an out-of-bounds read on the device.  CPU part: the checks raise before anything is launched."""
import pytest
import torch

import bitblas_amd as bitblas


def _linear(opt_M):
    return bitblas.Linear(256, 128, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True,
                          zeros_mode="original", opt_M=opt_M, enable_tuning=False)


def test_forward_refuses_cpu_tensors():
    lin = _linear([1, 16])
    with pytest.raises(RuntimeError, match="GPU only"):
        lin(torch.zeros(1, 256, dtype=torch.float16))
    with pytest.raises(RuntimeError, match="GPU only"):
        lin.bitblas_matmul(torch.zeros(1, 256, dtype=torch.float16), lin.qweight, lin.scales, lin.zeros)


@pytest.mark.gpu
def test_forward_checks_shape_dtype_and_static_m():
    lin = _linear(4).cuda()                       # static M = 4
    ok = torch.zeros(4, 256, dtype=torch.float16, device="cuda")
    assert lin(ok).shape == (4, 128)
    with pytest.raises(ValueError, match="built for M=4"):
        lin(torch.zeros(3, 256, dtype=torch.float16, device="cuda"))       # fewer rows: the kernel would read past A
    with pytest.raises(ValueError, match="built for M=4"):
        lin(torch.zeros(8, 256, dtype=torch.float16, device="cuda"))       # more rows: rows 4.. would stay unwritten
    with pytest.raises(ValueError, match="columns"):
        lin(torch.zeros(4, 128, dtype=torch.float16, device="cuda"))
    with pytest.raises(TypeError, match="A_dtype"):
        lin(torch.zeros(4, 256, dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError, match="output"):
        lin(ok, output=torch.zeros(128, 4, dtype=torch.float16, device="cuda").t())
    with pytest.raises(ValueError, match="output must hold"):
        lin(ok, output=torch.zeros(4, 64, dtype=torch.float16, device="cuda"))        # too small: an out-of-bounds write
    with pytest.raises(ValueError, match="output must hold"):
        lin(ok, output=torch.zeros(4, 128, dtype=torch.float32, device="cuda"))       # right shape, wrong element type
    with pytest.raises(ValueError, match="output must hold"):
        lin.bitblas_matmul(ok, lin.qweight, lin.scales, lin.zeros, output=torch.zeros(2, 128, dtype=torch.float16, device="cuda"))
    flat = torch.zeros(4 * 128, dtype=torch.float16, device="cuda")                   # same bytes, another view: fine
    assert lin(ok, output=flat).data_ptr() == flat.data_ptr()


@pytest.mark.gpu
def test_linear_follows_replaced_scale_zero_and_bias_buffers():
    """assigning `layer.scales` / `.zeros` / `.bias` after a forward must take effect (upstream re-reads the pointers
    on every call; the cached pointers here are compared with the live buffers)"""
    import numpy as np
    from helpers import assert_fp_parity, make_case, oracle_output
    case = make_case(2, 128, 256, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, with_bias=True, seed=5)
    lin = bitblas.Linear(256, 128, bias=True, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True,
                         with_zeros=True, zeros_mode="original", opt_M=[1, 16], enable_tuning=False).cuda()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()       # noqa: E731
    lin.load_and_transform_weight(dev(case["w_user"]), scales=dev(case["scale"]), zeros=dev(case["zeros"]), bias=dev(case["bias"]))
    A = dev(case["A"])
    assert_fp_parity(lin(A).cpu().numpy(), oracle_output(case))
    case2 = dict(case)
    case2["scale"] = (case["scale"].astype(np.float32) * 0.5).astype(np.float16)
    case2["zeros"] = (case["zeros"] + 1).astype(np.float16)
    case2["bias"] = (case["bias"] + 0.25).astype(np.float16)
    lin.scales = dev(case2["scale"])              # direct assignment, no load_and_transform_weight
    lin.zeros = dev(case2["zeros"])
    lin.bias = dev(case2["bias"])
    assert_fp_parity(lin(A).cpu().numpy(), oracle_output(case2))


@pytest.mark.gpu
def test_empty_batch_and_leading_batch_dimensions():
    """dynamic-M operator: `m = prod(A.shape[:-1])` (general_matmul/__init__.py:735-741); m == 0 returns at once like the
    generated dispatcher (`if (m == 0) return;`, builder/wrapper/tl.py:277) - no launch, an empty (0, N) result"""
    import numpy as np
    import torch
    import bitblas_amd as bitblas
    from helpers import assert_fp_parity, make_case, oracle_output
    case = make_case(6, 256, 512, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.05, seed=1)
    cfg = bitblas.MatmulConfig(M=[1, 16], N=256, K=512, A_dtype="float16", W_dtype="int4", group_size=128, with_scaling=True)
    mm = bitblas.Matmul(cfg, enable_tuning=False)
    W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()      # pre-offset codes, as the reference's op test feeds them
    scale = torch.from_numpy(case["scale"]).cuda()
    empty = mm(torch.empty((0, 512), dtype=torch.float16, device="cuda"), W, scale=scale)
    assert tuple(empty.shape) == (0, 256)
    torch.cuda.synchronize()
    A = torch.from_numpy(case["A"]).cuda().reshape(2, 3, 512)
    out = mm(A, W, scale=scale)
    assert tuple(out.shape) == (2, 3, 256)
    assert_fp_parity(out.reshape(6, 256).cpu().numpy(), oracle_output(case))
    # the C entry itself
    import ctypes
    from bitblas_amd import lib as wl
    st = wl.load_library().wqaa_matmul(ctypes.byref(mm.lib.desc), None, None, None, None, None, None, None, 0, None)
    assert st == 0


def test_forward_ex_and_gate_up_check_before_they_launch(monkeypatch):
    """`Matmul.forward_ex` / `matmul_gate_up` hand raw pointers to the C ABI like `forward`: a residual or output of the wrong size
    or type, or two projections of different configurations, are refused in Python (CPU tensors, device check stubbed)"""
    lin = _linear([1, 16])
    mm = lin.bitblas_matmul
    monkeypatch.setattr(mm, "check_activation", lambda A: A.numel() // A.shape[-1], raising=False)
    A = torch.zeros(1, 256, dtype=torch.float16)
    with pytest.raises(ValueError, match="residual"):
        mm.forward_ex(A, lin.qweight, lin.scales, lin.zeros, residual=torch.zeros(1, 64, dtype=torch.float16))
    with pytest.raises(ValueError, match="residual"):
        mm.forward_ex(A, lin.qweight, lin.scales, lin.zeros, residual=torch.zeros(1, 128, dtype=torch.float32))
    with pytest.raises(ValueError, match="output must hold"):
        mm.forward_ex(A, lin.qweight, lin.scales, lin.zeros, residual=torch.zeros(1, 128, dtype=torch.float16),
                      output=torch.zeros(1, 64, dtype=torch.float16))
    with pytest.raises(ValueError, match="bytes"):
        mm.forward_ex(A, lin.qweight[:64], lin.scales, lin.zeros, residual=torch.zeros(1, 128, dtype=torch.float16))
    other = bitblas.Linear(256, 64, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True,
                           zeros_mode="original", opt_M=[1, 16], enable_tuning=False).bitblas_matmul
    monkeypatch.setattr(other, "check_activation", lambda A: A.numel() // A.shape[-1], raising=False)
    with pytest.raises(ValueError, match="one configuration"):
        bitblas.matmul_gate_up(mm, other, A, lin.qweight, lin.qweight)
    with pytest.raises(ValueError, match="output must hold"):
        bitblas.matmul_gate_up(mm, mm, A, (lin.qweight, lin.scales, lin.zeros), (lin.qweight, lin.scales, lin.zeros),
                               output=torch.zeros(1, 64, dtype=torch.float16))
