"""`Linear.enable_decoded_weight_cache` (bitblas_amd/module.py): host logic on CPU with the two device calls replaced by
stand-ins; the device half is tests/test_decoded_cache_gpu.py."""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle


def make_linear(bias=False):
    lin = bitblas.Linear(256, 64, bias=bias, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True,
                         with_zeros=True, zeros_mode="original", opt_M=[1, 16, 512], enable_tuning=False)
    rng = np.random.default_rng(0)
    codes = rng.integers(0, 16, size=(64, 256)).astype(np.int8)
    lin.load_and_transform_weight(torch.from_numpy(codes),
                                  scales=torch.from_numpy((rng.random((64, 2)) * 0.05).astype(np.float16)),
                                  zeros=torch.full((64, 2), 8.0, dtype=torch.float16),
                                  bias=torch.from_numpy(rng.standard_normal(64).astype(np.float16)) if bias else None)
    return lin, codes


def cpu_stand_ins(lin, codes, monkeypatch):
    """replace the launches: B_decode from the oracle, the dense GEMM from torch, the packed path by a recorder"""
    calls = {"decode": 0, "dense": 0, "packed": 0}
    mm = lin.bitblas_matmul

    def fake_decode(W, scale=None, zeros=None, out=None):
        calls["decode"] += 1
        live = bitblas.lib.unpack_weight(W.numpy(), 256, 4, bitblas.lib.LAYOUT_LOP3 if mm.config.fast_decoding else bitblas.lib.LAYOUT_PLAIN,
                                         bitblas.lib.F16)
        d = oracle.dequantize_weight(live, "uint", 4, K=256, scale=scale.numpy(), zeros=zeros.numpy(), zeros_mode="original",
                                     group_size=128)
        t = torch.from_numpy(np.asarray(d, dtype=np.float16))
        if out is not None:
            out.copy_(t)
            return out
        return t

    def fake_dense(A, W, scale=None, zeros=None, bias=None, output=None):
        calls["dense"] += 1
        r = (A.float() @ W.float().t()).half()
        if output is not None:
            output.copy_(r)
            return output
        return r

    def fake_run(*a, **k):
        calls["packed"] += 1

    monkeypatch.setattr(mm, "check_activation", lambda A: A.numel() // A.shape[-1], raising=False)
    monkeypatch.setattr(mm, "dequantize_weight", fake_decode, raising=False)
    monkeypatch.setattr(mm.lib, "run", fake_run, raising=False)
    monkeypatch.setattr("bitblas_amd.module.current_stream_handle", lambda dev: 0)
    return calls, fake_dense


def test_enable_validates():
    dense = bitblas.Linear(256, 64, A_dtype="float16", W_dtype="float16", opt_M=[1, 16], enable_tuning=False)
    with pytest.raises(ValueError, match="already is its own"):
        dense.enable_decoded_weight_cache()
    i8 = bitblas.Linear(256, 64, A_dtype="int8", W_dtype="int2", accum_dtype="int32", out_dtype="int32", opt_M=[1, 16],
                        enable_tuning=False)
    with pytest.raises(ValueError, match="float16 / bfloat16"):
        i8.enable_decoded_weight_cache()
    lin, _ = make_linear()
    with pytest.raises(ValueError, match="min_m >= 16"):
        lin.enable_decoded_weight_cache(min_m=4)
    assert lin.enable_decoded_weight_cache(min_m=64) is lin
    cfg = lin._dense_op.config
    assert (cfg.A_dtype, cfg.W_dtype, cfg.N, cfg.K, cfg.with_bias) == ("float16", "float16", 64, 256, False)
    assert lin._dense_op.plans[512]["kernel_family"] in (2, 3)      # an MFMA or vendor-library dense member, never a decode
    lin.disable_decoded_weight_cache()
    assert lin._decoded_min_m == 0 and lin._dense_op is None


@pytest.mark.parametrize("bias", [False, True])
def test_dispatch_and_invalidation(bias, monkeypatch):
    lin, codes = make_linear(bias)
    lin.enable_decoded_weight_cache(min_m=64)
    calls, fake_dense = cpu_stand_ins(lin, codes, monkeypatch)
    monkeypatch.setattr(lin._dense_op, "forward", fake_dense, raising=False)
    rng = np.random.default_rng(1)
    A = torch.from_numpy((rng.random((100, 256)) - 0.5).astype(np.float16))
    out = lin(A)
    assert calls == {"decode": 1, "dense": 1, "packed": 0}
    d = oracle.dequantize_weight(codes, "uint", 4, K=256, scale=lin.scales.numpy(), zeros=lin.zeros.numpy(),
                                 zeros_mode="original", group_size=128)
    want = (A.float() @ torch.from_numpy(np.asarray(d, dtype=np.float32)).t()).half()
    if bias:
        want = want + lin.bias
    assert torch.equal(out, want)
    lin(A)
    assert calls["decode"] == 1 and calls["dense"] == 2              # resident: not decoded again
    lin(A[:8])                                                        # below the threshold: the packed path
    assert calls["packed"] == 1 and calls["dense"] == 2
    lin.scales.mul_(2)                                                # in place (what load_state_dict does): re-decoded
    out2 = lin(A)
    assert calls["decode"] == 2
    assert not torch.equal(out2, out)
    new_codes = torch.from_numpy(((codes.astype(np.int16) + 1) % 16).astype(np.int8))
    lin.load_and_transform_weight(new_codes, scales=lin.scales, zeros=lin.zeros, bias=lin.bias)      # replaced tensors
    lin(A)
    assert calls["decode"] == 3
    sd = {k: v.clone() for k, v in lin.state_dict().items()}
    lin.load_state_dict(sd)                                           # copy_ into the live buffers: versions move
    lin(A)
    assert calls["decode"] == 4
