"""BitNet caller ops fused at the boundary: HIP quantiser + fused epilogue
vs the oracle restatement of integration/BitNet/utils_quant.py:150-216."""
import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
from bitblas_amd.bitnet import BitLinear

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,K", [(1, 4096), (7, 1024), (300, 2560)])
def test_activation_quant_matches_reference_math(rows, K):
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, K)) * 3).astype(np.float16)
    x[0, :8] = 0
    lin = BitLinear(K, 256).cuda()
    q, s = lin.activation_quant(torch.from_numpy(x).cuda())
    wq, ws = oracle.bitnet_activation_quant(x)
    assert np.array_equal(q.cpu().numpy(), wq)
    assert np.array_equal(s.cpu().numpy(), ws[:, 0])


@pytest.mark.parametrize("m", [1, 5, 16, 64, 300])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("seed", [0, 100])
def test_bitlinear_forward(m, bias, seed):
    rng = np.random.default_rng(m + seed + 100 * bias)
    N, K = 512, 1024
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float16) if bias else None
    lin = BitLinear(K, N, bias=bias).cuda()
    lin.load_float_weight(torch.from_numpy(w).cuda(), None if b is None else torch.from_numpy(b).cuda())
    x = (rng.standard_normal((m, K))).astype(np.float16)
    got = lin(torch.from_numpy(x).cuda()).cpu().numpy()
    # sw / the ternary codes come from the module (torch's reduction order for mean|W| differs from
    # numpy's in the last bit); what is under test is quantiser + matmul + fused epilogue
    wq = BitLinear.weight_quant(torch.from_numpy(w)).numpy()
    want = oracle.bitnet_forward(x, wq, np.float32(lin.sw.item()), b)
    # same integer accumulators, same two fp32 divisions, same half rounding: bit exact
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_bitlinear_llama_shape_exact():
    rng = np.random.default_rng(0)
    N, K = 4096, 4096
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    lin = BitLinear(K, N).cuda()
    lin.load_float_weight(torch.from_numpy(w).cuda())
    x = rng.standard_normal((2, K)).astype(np.float16)
    got = lin(torch.from_numpy(x).cuda()).cpu().numpy()
    wq = BitLinear.weight_quant(torch.from_numpy(w)).numpy()
    assert np.array_equal(got.view(np.uint16), oracle.bitnet_forward(x, wq, np.float32(lin.sw.item())).view(np.uint16))
    wq_np, sw_np = oracle.bitnet_weight_quant(w)
    assert np.array_equal(wq_np, wq) and abs(float(sw_np) - lin.sw.item()) <= 1e-6 * float(sw_np)
