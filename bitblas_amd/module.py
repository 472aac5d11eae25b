"""`bitblas.Linear` for MI355X: an nn.Module whose forward is one HIP kernel launch.

Interface and buffer layout follow bitblas/module/__init__.py:77-367 (buffers `qweight`, `scales`,
`zeros`, `bias`; `load_and_transform_weight`; `repack_from_gptq[_v2]`; `opt_M` dynamic ranges) so
checkpoints and calling code carry over.  The qweight bytes are the reference's own
`transform_weight` layout, so a state_dict saved by upstream BitBLAS loads unchanged.
"""
from __future__ import annotations

import ctypes
import operator
from functools import reduce
from logging import getLogger
from typing import List, Optional, Union

import torch
import torch.nn as nn

from .cache import get_database_path, global_operator_cache
from .lib import current_stream_handle
from .matmul import Matmul, MatmulConfig, torch_dtype
from .quantization import general_compress
from .target import auto_detect_nvidia_target

logger = getLogger(__name__)


def _unpack_fields(packed: torch.Tensor, word_dtype: torch.dtype, bits: int) -> torch.Tensor:
    """Expand every `word_dtype` word of `packed` into its bit fields (lowest first), truncated to
    int8 like the reference's loops (module/__init__.py:24-74)."""
    words = packed.view(word_dtype)
    per_word = (torch.iinfo(word_dtype).bits) // bits
    shifts = torch.arange(per_word, device=words.device, dtype=word_dtype) * bits
    fields = (words.unsqueeze(-1) >> shifts).to(torch.int8)
    return fields.reshape(words.shape[0], words.shape[1] * per_word)


def unpack_qzeros(qzeros, bits):
    """GPTQ (v1) zero points: stored minus one (AutoGPTQ qlinear_cuda_old), :24-39."""
    return torch.bitwise_and(_unpack_fields(qzeros, torch.int32, bits) + 1, 2 ** bits - 1)


def unpack_qzeros_v2(qzeros, bits):
    """GPTQModel v2 zero points: stored as-is, :43-58."""
    return torch.bitwise_and(_unpack_fields(qzeros, torch.int32, bits), 2 ** bits - 1)


def unpack_qweight(qweight, bits):
    """:61-74"""
    return torch.bitwise_and(_unpack_fields(qweight, torch.int8, bits), 2 ** bits - 1)


class Linear(nn.Module):
    opt_M = [16, 32, 64, 128, 256, 512]
    STORAGE_DTYPE = "int8"
    TORCH_STORAGE_DTYPE = torch.int8
    BITBLAS_DTYPES = {torch.float32: "float32", torch.float16: "float16", torch.half: "float16",
                      torch.int8: "int8"}

    def __init__(self, in_features: int, out_features: int, bias: bool = False,
                 A_dtype: str = "float16", W_dtype: str = "float16", accum_dtype: str = "float16",
                 out_dtype: str = "float16", group_size: int = -1, with_scaling: bool = None,
                 with_zeros: bool = False, zeros_mode: str = None,
                 opt_M: Union[int, List[int]] = opt_M, enable_tuning: bool = True,
                 fast_decoding: Optional[bool] = None, propagate_b: bool = False):
        """`opt_M`: an int builds a static-M operator, a list a dynamic-M one (one kernel choice
        per bucket, `m <= opt` dispatch like the reference's generated wrapper)."""
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.opt_M = opt_M
        self.group_size = in_features if group_size in (-1, None) else group_size
        self.torch_dtype = torch_dtype(A_dtype)
        self.is_consitent = A_dtype == W_dtype  # sic
        self.zeros_mode = zeros_mode
        if in_features % 16 != 0 or out_features % 16 != 0:
            raise ValueError("`in_features` and `out_features` must be divisible by 16.")
        if in_features % self.group_size != 0:
            raise ValueError("`in_features` must be divisible by `group_size`.")
        config = MatmulConfig(
            M=self.opt_M, N=out_features, K=in_features, A_dtype=A_dtype, W_dtype=W_dtype,
            accum_dtype=accum_dtype, out_dtype=out_dtype, storage_dtype=self.STORAGE_DTYPE,
            with_scaling=with_scaling, with_zeros=with_zeros, group_size=self.group_size,
            fast_decoding=fast_decoding, with_bias=bias, propagate_b=propagate_b,
            zeros_mode=zeros_mode)
        self.bitblas_matmul = self._get_or_create_bitblas_operator(config, enable_tuning)
        self.bits = self.bitblas_matmul.bit
        self.source_format = self.bitblas_matmul.source_format
        self._initialize_buffers(in_features, out_features, bias)
        self.q_params = None
        self._decoded_min_m = 0          # enable_decoded_weight_cache
        self._decoded = self._decoded_key = self._dense_op = None

    @property
    def consistent(self):
        return self.is_consitent

    def _initialize_buffers(self, in_features, out_features, bias):
        groups = in_features // self.group_size
        if self.consistent:
            # upstream allocates (out, in // group_size) here; the dense weight is (out, in)
            self.register_buffer("weight", torch.zeros((out_features, in_features), dtype=self.torch_dtype))
        else:
            self.register_buffer("qweight", torch.zeros(self.bitblas_matmul.retrieve_weight_shape(),
                                                        dtype=self.TORCH_STORAGE_DTYPE))
            self.register_buffer("scales", torch.zeros((out_features, groups), dtype=self.torch_dtype))
            if self.zeros_mode == "quantized":
                self.register_buffer("zeros", torch.zeros((groups, out_features // 8 * self.bits),
                                                          dtype=self.TORCH_STORAGE_DTYPE))
            else:
                self.register_buffer("zeros", torch.zeros((out_features, groups), dtype=self.torch_dtype))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features,), dtype=self.torch_dtype))
        else:
            self.bias = None

    def _get_or_create_bitblas_operator(self, config, enable_tuning):
        target = auto_detect_nvidia_target()
        if global_operator_cache.size() == 0:
            global_operator_cache.load_from_database(get_database_path(), target)
            logger.info("Loaded %d operators from database.", global_operator_cache.size())
        op = global_operator_cache.get(config)
        if op is None:
            op = Matmul(config, target=target, enable_tuning=False)
            if enable_tuning:
                op.hardware_aware_finetune(topk=20)
            global_operator_cache.add(config, op)
            logger.info("BitBLAS operator created: %s", op.get_kernel_name_generator().generate())
        else:
            logger.info("BitBLAS operator found in global_operator_cache.")
        return op

    def warmup(self, topk=20):
        self.bitblas_matmul.hardware_aware_finetune(topk=topk)

    def _live_params(self):
        """the buffers the kernel reads, in `lib.call` order (B, scale, zeros, bias)"""
        cfg = self.bitblas_matmul.config
        if self.consistent:
            params = [self.weight]
        else:
            params = [self.qweight]
            if cfg.with_scaling:
                params.append(self.scales)
            if cfg.with_zeros:
                params.append(self.zeros)
        if cfg.with_bias:
            params.append(self.bias)
        return params

    def init_params(self):
        """Pre-wrap the parameter pointers (upstream redoes this on every forward, :138-153)."""
        cfg = self.bitblas_matmul.config
        params = self._live_params()
        self.q_params = [ctypes.c_void_p(p.data_ptr()) for p in params]
        self._q_param_keys = tuple(p.data_ptr() for p in params)
        # raw pointers in `BoundLib.run` order (B, scale, zeros, bias) for the forward fast path
        it = iter(self._q_param_keys[1:])
        self._q_run = (self._q_param_keys[0],
                       next(it) if (not self.consistent and cfg.with_scaling) else None,
                       next(it) if (not self.consistent and cfg.with_zeros) else None,
                       next(it) if cfg.with_bias else None)

    def _params_current(self):
        """the cached pointers still are the LIVE buffers' (`layer.scales = t`, `.to(device)`, `load_state_dict`
        all replace tensors; upstream re-reads every pointer on every forward)"""
        if self.q_params is None:
            return False
        live = self._live_params()
        return len(live) == len(self._q_param_keys) and all(t.data_ptr() == k for t, k in zip(live, self._q_param_keys))

    # -- MI355X extension: the decoded weight kept resident for large-M calls ----------------------------------------------
    def enable_decoded_weight_cache(self, min_m: int = 256):
        """From `min_m` activation rows on, multiply by a RESIDENT copy of `B_decode` - the TE graph's first stage
        (tirscript/matmul_dequantize_impl.py:391-449), written once by `Matmul.dequantize_weight` - through the plain dense
        GEMM of the same shape, instead of decoding the packed weight again in every call.  It is the two-pass member
        (`wqaa_matmul_desc.two_pass_min_m`) with its first pass hoisted out of the call: same B_decode bits, same GEMM, so
        the same results; what it costs is memory, N*K*sizeof(A_dtype) per layer (4x a 4-bit weight) - the trade 288 GB of
        HBM per GPU is there for: a 70B model's float16 B_decode is 140 GB next to 35 GB of packed weights.  Prefill
        (M in the thousands) takes the dense path, decode (M = 1 ... 64) keeps streaming 4-bit weights.
        The copy follows the live buffers: replacing `qweight` / `scales` / `zeros`, writing into them in place
        (`load_state_dict`) or moving the module re-decodes at the next large-M call.  float16 / bfloat16 activations."""
        cfg = self.bitblas_matmul.config
        if self.consistent:
            raise ValueError("W_dtype == A_dtype: the weight already is its own B_decode")
        if cfg.A_dtype not in ("float16", "bfloat16") or cfg.K % 128 != 0:
            raise ValueError("the decoded-weight cache serves float16 / bfloat16 activations with K a multiple of 128")
        if int(min_m) < 16:
            raise ValueError("min_m >= 16: below that the packed weight stream IS the cost (GEMV / decode-batch members)")
        dense = MatmulConfig(M=cfg.M, N=cfg.N, K=cfg.K, A_dtype=cfg.A_dtype, W_dtype=cfg.A_dtype, accum_dtype=cfg.accum_dtype,
                             out_dtype=cfg.out_dtype, with_bias=False)
        self._dense_op = Matmul(dense, enable_tuning=False)
        self._decoded_min_m = int(min_m)
        self._decoded = self._decoded_key = None
        return self

    def disable_decoded_weight_cache(self):
        self._decoded_min_m = 0
        self._decoded = self._decoded_key = self._dense_op = None

    def _decoded_weight(self):
        cfg = self.bitblas_matmul.config
        live = [self.qweight] + ([self.scales] if cfg.with_scaling else []) + ([self.zeros] if cfg.with_zeros else [])
        key = tuple((t.data_ptr(), t._version) for t in live)
        if self._decoded is None or key != self._decoded_key or self._decoded.device != self.qweight.device:
            self._decoded = self.bitblas_matmul.dequantize_weight(
                self.qweight, self.scales if cfg.with_scaling else None, self.zeros if cfg.with_zeros else None,
                out=self._decoded if self._decoded is not None and self._decoded.device == self.qweight.device else None)
            self._decoded_key = key
        return self._decoded

    def forward(self, A, output=None):
        mm = self.bitblas_matmul
        A = mm.transform_input(A)
        m = mm.check_activation(A)           # cuda, K columns, A_dtype, and M rows for a static-M operator
        if self._decoded_min_m and m >= self._decoded_min_m:
            out = self._dense_op.forward(A, self._decoded_weight(), output=output)
            if self.bias is not None:
                out += self.bias             # after the cast to out_dtype, as the TE graph adds it (:462-477)
            return out
        if not A.is_contiguous():
            A = A.contiguous()   # the kernels read raw row-major memory
        if not self._params_current():
            self.init_params()
        if output is None:
            # upstream uses torch.zeros here; every element is written by the kernel
            output = torch.empty(A.shape[:-1] + (self.out_features,),
                                 dtype=torch_dtype(mm.out_dtype), device=A.device)
        elif not output.is_contiguous() or output.device != A.device:
            raise ValueError("output must be a contiguous tensor on A's device")
        else:
            mm.check_output(output, m)
        # upstream rebuilds a ctypes argument list and goes through the positional `lib.call` here
        # (:271-287); same pointers, same order, without the per-call wrapping
        lut = mm._ensure_lut(A.device) if self.source_format == "nf" else None
        B, scale, zeros, bias = self._q_run
        mm.lib.run(A.data_ptr(), B, lut.data_ptr() if lut is not None else None, scale, zeros, bias,
                   output.data_ptr(), m, current_stream_handle(A.device), A.device)
        return output

    def forward_ex(self, A, residual=None, output=None, norm=None):
        """`residual + forward(A)` - one launch at decode row counts (`Matmul.forward_ex`): what a decoder layer writes as
        `hidden = residual + o_proj(attn)` / `residual + down_proj(act)`; or, `norm` = (weight, eps), `forward(rms_norm(A))`."""
        if residual is None and norm is None:
            return self.forward(A, output=output)
        if not self._params_current():
            self.init_params()
        cfg = self.bitblas_matmul.config
        quantised = not self.consistent
        A = self.bitblas_matmul.transform_input(A)          # as `forward` does (module/__init__.py:271-276)
        return self.bitblas_matmul.forward_ex(A, self.qweight if quantised else self.weight,
                                              self.scales if quantised and cfg.with_scaling else None,
                                              self.zeros if quantised and cfg.with_zeros else None,
                                              self.bias if cfg.with_bias else None, output, residual=residual, norm=norm)

    def load_and_transform_weight(self, weight: torch.Tensor, scales: torch.Tensor = None,
                                  zeros: torch.Tensor = None, bias: torch.Tensor = None):
        if self.consistent:
            assert scales is None, "scales should be None for consistent mode."
            assert zeros is None, "zeros should be None for consistent mode."
            self.weight = self.bitblas_matmul.transform_weight(weight)
        else:
            self.qweight = self.bitblas_matmul.transform_weight(weight)
            if scales is not None:
                self.scales = scales
            if zeros is not None:
                self.zeros = zeros
        if bias is not None:
            self.bias = bias
        self.q_params = None

    def _repack(self, gptq_module, intzeros, device):
        qweight = gptq_module.qweight.T.contiguous().view(self.TORCH_STORAGE_DTYPE)
        mm = self.bitblas_matmul
        if mm.weight_transform is not None:
            ops = list(mm.weight_transform.operators)
            if ops and ops[0] is mm.weight_compress and self.bits in (1, 2, 4):
                # a transposed GPTQ word holds its fields lowest first, which IS the general_compress order: the reference's
                # unpack -> compress round trip (:315-338, ops/quant_compress) returns these bytes - only the later stages (the
                # LOP3 interleave) have anything to do.  (11008 x 4096: ~250 -> ~20 ms per layer on 8 host cores; the unpack alone was 226 ms.)
                out = qweight.cpu()
                for op in ops[1:]:
                    out = op.forward(out)
                qweight = out.to(device)
            else:
                intweight = unpack_qweight(qweight, self.bits).contiguous()
                qweight = mm.weight_transform(intweight.cpu()).to(device)
        self.qweight = qweight
        self.scales = gptq_module.scales.T.contiguous().view(self.torch_dtype).to(device)
        mode = self.bitblas_matmul.config.zeros_mode
        intzeros = intzeros.T.contiguous()
        if mode == "original":
            self.zeros = intzeros.to(torch.float16).contiguous().to(device)
        elif mode == "rescale":
            self.zeros = (intzeros.to(torch.float16).to(device) * self.scales).contiguous()
        elif mode == "quantized":
            packed = general_compress(intzeros.T.contiguous().cpu().numpy(), self.bits)
            self.zeros = torch.from_numpy(packed).to(device).to(self.zeros.dtype).contiguous()
        else:
            raise ValueError(f"Unsupported zeros type: {mode}")
        if self.bias is not None:
            self.bias = gptq_module.bias.data.to(torch.float16).contiguous().to(device)
        self.q_params = None

    def repack_from_gptq(self, gptq_module, device="cuda"):
        """AutoGPTQ `CudaOldQuantLinear` -> BitBLAS buffers (:315-338): qweight (in/8*bits, out)
        int32 is transposed and re-packed, scales transposed, zero points are +1-corrected."""
        self._repack(gptq_module, unpack_qzeros(gptq_module.qzeros, self.bits), device)

    def repack_from_gptq_v2(self, gptq_module, device="cuda"):
        """GPTQModel v2 checkpoints: zero points stored without the -1 offset (:340-363)."""
        self._repack(gptq_module, unpack_qzeros_v2(gptq_module.qzeros, self.bits), device)


__all__ = ["Linear", "unpack_qzeros", "unpack_qzeros_v2", "unpack_qweight"]
