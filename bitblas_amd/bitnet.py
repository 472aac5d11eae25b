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

    @classmethod
    def from_bit_linear(cls, bitlinear, weight_group: int = 1, opt_M=None):
        """`BitLinearBitBLAS.from_bit_linear` (utils_quant.py:103-114): a float `BitLinear` / `nn.Linear` -> this layer, ternary
        weights packed in the reference layout.  `weight_group` > 1 is how upstream gives the q / k / v blocks of a
        concatenated weight their own `sw` (:116-138); here the projections stay separate layers with a scalar `sw` each and
        share a launch through `BitLinearGroup`, so only 1 is accepted."""
        if weight_group != 1:
            raise NotImplementedError("weight_group > 1: keep q / k / v as separate layers and wrap them in BitLinearGroup")
        layer = cls(bitlinear.in_features, bitlinear.out_features, bias=bitlinear.bias is not None, opt_M=opt_M)
        layer.load_float_weight(bitlinear.weight.data, None if bitlinear.bias is None else bitlinear.bias.data)
        return layer

    def activation_quant(self, x: torch.Tensor):
        """utils_quant.py:157-164 on the GPU: one HIP launch."""
        x = x.contiguous()
        if x.dtype != torch.float16:
            x = x.to(torch.float16)
        q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
        s = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        _lib.act_quant_int8(x, q, s, _lib.current_stream_handle(x.device))
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
                                                    _lib.current_stream_handle(x.device), sw)
            return out
        q, si = self.activation_quant(x)
        self.bitblas_matmul.lib.run_fused(q.data_ptr(), self.qweight.data_ptr(),
                                          None if self.bias is None else self.bias.data_ptr(), out.data_ptr(), m,
                                          _lib.current_stream_handle(x.device), si.data_ptr(), sw)
        return out


class BitLinearGroup(nn.Module):
    """`BitLinear` layers that read the same input (q/k/v, gate/up of a BitNet block), called as one:
    `q, k, v = BitLinearGroup([q_proj, k_proj, v_proj])(x)`.

    Decode steps (m <= 2) run the whole group - per-token activation quantiser, W_int2 x A_int8 matmuls, `out / si / sw ->
    half (+bias)` - as ONE launch (`wqaa_matmul_group_ex`); larger batches quantise the activations once and run the layers
    one by one on the MFMA members.  Results are bit-identical to the layers' own `forward`.  The reference fuses the same
    projections by concatenating their float weights before quantisation (integration/BitNet/modeling_bitnet.py:1433-1445),
    which also merges their weight scales `sw`; here every layer keeps its own `sw` and packed tensor."""

    def __init__(self, layers):
        super().__init__()
        from .group import GROUP_MAX
        if not 1 <= len(layers) <= GROUP_MAX:
            raise ValueError(f"a group holds 1..{GROUP_MAX} layers")
        if len({l.in_features for l in layers}) != 1:
            raise ValueError("the layers of a group take the same input")
        self.layers = nn.ModuleList(layers)

    def forward(self, x: torch.Tensor):
        import ctypes
        from .group import GroupItem, _library
        layers = list(self.layers)
        l0 = layers[0]
        if not x.is_cuda:
            raise RuntimeError("bitblas_amd.bitnet.BitLinearGroup runs on the GPU only")
        m = x.numel() // l0.in_features
        if not (m <= 2 and x.dtype == torch.float16 and all(l.fuse_activation_quant for l in layers)):
            if m <= 4 or x.dtype != torch.float16:
                return tuple(l(x) for l in layers)
            # larger batches: one quantiser launch for the group, then the layers' matmuls with the fused post-process
            q, si = l0.activation_quant(x)
            outs = []
            stream = _lib.current_stream_handle(x.device)
            for l in layers:
                out = torch.empty(x.shape[:-1] + (l.out_features,), dtype=torch.float16, device=x.device)
                sw = getattr(l, "_sw_host", None)
                if sw is None:
                    sw = l._sw_host = float(l.sw)
                l.bitblas_matmul.lib.run_fused(q.data_ptr(), l.qweight.data_ptr(), None if l.bias is None else l.bias.data_ptr(),
                                               out.data_ptr(), m, stream, si.data_ptr(), sw)
                outs.append(out)
            return tuple(outs)
        xc = x if x.is_contiguous() else x.contiguous()
        n = len(layers)
        items = (GroupItem * n)()
        epis = (_lib.Epilogue * n)()
        eptr = (ctypes.POINTER(_lib.Epilogue) * n)()
        outs = []
        for i, l in enumerate(layers):
            out = torch.empty(x.shape[:-1] + (l.out_features,), dtype=torch.float16, device=x.device)
            outs.append(out)
            sw = getattr(l, "_sw_host", None)
            if sw is None:
                sw = l._sw_host = float(l.sw)
            it = items[i]
            it.desc = ctypes.pointer(l.bitblas_matmul.lib.desc)
            it.A, it.B, it.C = xc.data_ptr(), l.qweight.data_ptr(), out.data_ptr()
            it.Bias = None if l.bias is None else l.bias.data_ptr()
            e = epis[i]
            e.struct_size = ctypes.sizeof(_lib.Epilogue)
            e.flags = _lib.EPI_QUANTIZE_INPUT
            e.row_scale = None
            e.tensor_scale = float(sw)
            eptr[i] = ctypes.pointer(e)
        lib = _library()
        status = lib.wqaa_matmul_group_ex(items, eptr, n, m, _lib.current_stream_handle(x.device))
        if status != _lib.OK:
            _lib.check(status)
        return tuple(outs)
