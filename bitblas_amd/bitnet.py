"""BitNet-b1.58 style linear on MI355X: the callers either side of the W_int2 x A_int8 hot path.

Mirrors `BitLinearBitBLAS` of the reference (integration/BitNet/utils_quant.py:37-219): ternary
weights (`weight_quant`, :150-155) stored as int2, activations quantised per token to int8
(`activation_quant`, :157-164), `out / si / sw -> half (+bias)` afterwards (`post_quant_process`,
:166-171).  Upstream runs the two wrappers as separate torch.compile kernels around a ~1 us GEMV;
here the quantiser is one HIP kernel (`wqaa_act_quant_int8`) and the post-process is folded into the
matmul epilogue (`wqaa_matmul_ex`): 2 launches instead of 3+, no fp32 intermediate in HBM.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import lib as _lib
from .matmul import Matmul, MatmulConfig


class BitLinear(nn.Module):
    opt_M = [1, 16, 32, 64, 128, 256, 512]

    def __init__(self, in_features: int, out_features: int, bias: bool = False, opt_M=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        config = MatmulConfig(M=opt_M or self.opt_M, N=out_features, K=in_features, A_dtype="int8", W_dtype="int2",
                              out_dtype="float16", accum_dtype="int32", with_bias=bias, with_scaling=False,
                              with_zeros=False, zeros_mode=None)
        self.bitblas_matmul = Matmul(config, enable_tuning=False)
        self.fuse_activation_quant = True      # batches of <= 4 rows: quantise inside the matmul launch
        self.register_buffer("qweight", torch.zeros(self.bitblas_matmul.retrieve_weight_shape(), dtype=torch.int8))
        self.register_buffer("sw", torch.ones((), dtype=torch.float32))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float16))
        else:
            self.bias = None

    @staticmethod
    def weight_quant(weight: torch.Tensor) -> torch.Tensor:
        """utils_quant.py:150-155: ternary {-1, 0, 1} with s = 1 / mean|W|."""
        weight = weight.float()
        s = 1 / weight.abs().mean().clamp(min=1e-5)
        return (weight * s).round().clamp(-1, 1).type(torch.int8)

    def load_float_weight(self, weight: torch.Tensor, bias: torch.Tensor = None):
        """`post_process_weights` (:140-148): keep sw, store the int2 operand in the reference layout."""
        self.sw = (1 / weight.float().abs().mean().clamp(min=1e-5)).to(torch.float32).to(self.sw.device)
        q = self.weight_quant(weight)
        self.qweight = self.bitblas_matmul.transform_weight(q).to(self.qweight.device)
        if bias is not None:
            self.bias = bias.to(torch.float16).to(self.qweight.device)
        self._sw_host = float(self.sw)

    def activation_quant(self, x: torch.Tensor):
        """utils_quant.py:157-164 on the GPU: one HIP launch."""
        x = x.contiguous()
        if x.dtype != torch.float16:
            x = x.to(torch.float16)
        q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
        s = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        _lib.act_quant_int8(x, q, s, torch.cuda.current_stream(x.device).cuda_stream)
        return q, s

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("bitblas_amd.bitnet.BitLinear runs on the GPU only")
        m = x.numel() // self.in_features
        out = torch.empty(x.shape[:-1] + (self.out_features,), dtype=torch.float16, device=x.device)
        sw = getattr(self, "_sw_host", None)
        if sw is None:
            sw = self._sw_host = float(self.sw)
        if m <= 4 and x.dtype == torch.float16 and self.fuse_activation_quant:
            # decode steps: quantise + matmul + rescale in ONE launch (the GEMV workgroup quantises the row itself)
            xc = x if x.is_contiguous() else x.contiguous()
            self.bitblas_matmul.lib.run_fused_quant(xc.data_ptr(), self.qweight.data_ptr(),
                                                    None if self.bias is None else self.bias.data_ptr(), out.data_ptr(), m,
                                                    torch.cuda.current_stream(x.device).cuda_stream, sw)
            return out
        q, si = self.activation_quant(x)
        self.bitblas_matmul.lib.run_fused(q.data_ptr(), self.qweight.data_ptr(),
                                          None if self.bias is None else self.bias.data_ptr(), out.data_ptr(), m,
                                          torch.cuda.current_stream(x.device).cuda_stream, si.data_ptr(), sw)
        return out
