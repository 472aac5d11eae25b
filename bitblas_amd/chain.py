"""Chains of DEPENDENT operators described once: `matmul_chain`, `DecoderTail`.

The reference runs a decoder layer as one operator call per `nn.Linear` with the caller's elementwise kernels between them
(integration/BitNet/modeling_bitnet.py: `BitnetMLP.forward` :240-244, `BitnetDecoderLayer.forward` :839-860).
`wqaa_matmul_chain` (include/wqaa.h) takes o_proj (+ residual) -> RMSNorm -> gate / up * silu -> down_proj (+ residual) - or
any chain of operators wired output -> input - and runs the launches it stands for, in order, each with the callers'
elementwise ops folded in (`Matmul.forward_ex`, `matmul_gate_up`).  (Rounds 3-5 also had a persistent one-launch member behind
the same call; bit-identical, 50 % slower, removed in round 6: docs/DESIGN_r04.md section 3.3c.)
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import lib as _lib
from .matmul import Matmul

CHAIN_MAX = 8


class ChainItem(ctypes.Structure):
    """struct wqaa_chain_item (include/wqaa.h)."""
    _fields_ = [("desc", ctypes.POINTER(_lib.MatmulDesc)), ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("Scale", ctypes.c_void_p),
                ("Zeros", ctypes.c_void_p), ("Bias", ctypes.c_void_p), ("B2", ctypes.c_void_p), ("Scale2", ctypes.c_void_p),
                ("Zeros2", ctypes.c_void_p), ("Bias2", ctypes.c_void_p), ("C", ctypes.c_void_p), ("residual", ctypes.c_void_p),
                ("norm_weight", ctypes.c_void_p), ("norm_eps", ctypes.c_float), ("kind", ctypes.c_int32),
                ("input_from", ctypes.c_int32), ("residual_from", ctypes.c_int32)]


_bound = False


def _library():
    global _bound
    lib = _lib.load_library()
    if not _bound:
        lib.wqaa_matmul_chain.restype = ctypes.c_int
        lib.wqaa_matmul_chain.argtypes = [ctypes.POINTER(ChainItem), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.wqaa_chain_plan.restype = ctypes.c_int
        lib.wqaa_chain_plan.argtypes = [ctypes.POINTER(ChainItem), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(_lib.Plan)]
        _bound = True
    return lib


Weights = Union[torch.Tensor, Tuple]


@dataclass
class ChainStep:
    """One item of a chain.

    op            the operator (`Matmul`, built for M = the chain's row count); `up_op` / `up_weights` make it the gate / up pair
                  `silu(op(x)) * up_op(x)` (`matmul_gate_up`)
    weights       `W` or `(W, scale, zeros, bias)`, as `Matmul.forward` after A
    input         a tensor (m, K), or the index of the earlier step whose output it reads
    residual      None, a tensor (m, N), or the index of an earlier step whose output is added (`Matmul.forward_ex(residual=)`)
    norm          None or (weight, eps): RMSNorm in front of the operator (`forward_ex(norm=)`)
    output        None: allocated and returned; a tensor: written; False: a temporary only later steps read (not returned)"""
    op: Matmul
    weights: Weights
    input: Union[torch.Tensor, int]
    residual: Union[None, torch.Tensor, int] = None
    norm: Optional[Tuple[torch.Tensor, float]] = None
    output: Union[None, torch.Tensor, bool] = None
    up_op: Optional[Matmul] = None
    up_weights: Optional[Weights] = None


def _w4(w, op, what):
    w = (w,) if isinstance(w, torch.Tensor) else tuple(w)
    W, scale, zeros, bias = (w + (None,) * 4)[:4]
    if W.numel() * W.element_size() != op._w_bytes:
        raise ValueError(f"{what}: W holds {W.numel() * W.element_size()} bytes, the operator expects {op._w_bytes} "
                         f"(shape {op.retrieve_weight_shape()}: run transform_weight first)")
    return W, scale, zeros, bias


def _build(steps: Sequence[ChainStep], allocate: bool):
    n = len(steps)
    if n == 0 or n > CHAIN_MAX:
        raise ValueError(f"a chain has 1..{CHAIN_MAX} steps (got {n})")
    items = (ChainItem * n)()
    outs: List[Optional[torch.Tensor]] = [None] * n
    keep = []
    m = None
    dev = None
    for i, st in enumerate(steps):
        op = st.op
        it = items[i]
        it.desc = ctypes.pointer(op.lib.desc)
        it.kind = 1 if st.up_op is not None else 0
        if st.up_op is not None and bytes(st.up_op.lib.desc) != bytes(op.lib.desc):
            raise ValueError(f"step {i}: gate and up must be operators of one configuration")
        if isinstance(st.input, torch.Tensor):
            a = st.input
            if allocate:
                mi = op.check_activation(a)
            else:                                   # planning reads shapes only (tensors may live on the CPU)
                if a.shape[-1] != op.K:
                    raise ValueError(f"step {i}: A has {a.shape[-1]} columns, the operator was built for K={op.K}")
                mi = a.numel() // a.shape[-1]
            if not a.is_contiguous():
                a = a.contiguous()
            keep.append(a)
            it.A, it.input_from = a.data_ptr(), -1
            dev = a.device if dev is None else dev
            if a.device != dev:
                raise ValueError("all tensors of a chain live on one device")
            m = mi if m is None else m
            if mi != m:
                raise ValueError(f"step {i}: {mi} activation rows, the chain has {m}")
        else:
            j = int(st.input)
            if not 0 <= j < i:
                raise ValueError(f"step {i}: input {j} is not an earlier step")
            if steps[j].op.N != op.K:
                raise ValueError(f"step {i} (K = {op.K}) reads step {j} (N = {steps[j].op.N})")
            it.A, it.input_from = None, j
        W, scale, zeros, bias = _w4(st.weights, op, f"step {i}")
        it.B = W.data_ptr()
        it.Scale = scale.data_ptr() if scale is not None else None
        it.Zeros = zeros.data_ptr() if zeros is not None else None
        it.Bias = bias.data_ptr() if bias is not None else None
        if st.up_op is not None:
            W2, s2, z2, b2 = _w4(st.up_weights, st.up_op, f"step {i} (up)")
            it.B2 = W2.data_ptr()
            it.Scale2 = s2.data_ptr() if s2 is not None else None
            it.Zeros2 = z2.data_ptr() if z2 is not None else None
            it.Bias2 = b2.data_ptr() if b2 is not None else None
        it.residual_from = -1
        if isinstance(st.residual, torch.Tensor):
            r = st.residual
            if r.dtype != torch.float16 or not r.is_contiguous() or r.shape[-1] != op.N:
                raise ValueError(f"step {i}: the residual must be a contiguous float16 tensor of width {op.N}")
            keep.append(r)
            it.residual = r.data_ptr()
        elif st.residual is not None:
            j = int(st.residual)
            if not 0 <= j < i or steps[j].op.N != op.N:
                raise ValueError(f"step {i}: residual {j} is not an earlier step of width {op.N}")
            it.residual_from = j
        if st.norm is not None:
            w, eps = st.norm
            if w.dtype != torch.float16 or not w.is_contiguous() or w.numel() != op.K:
                raise ValueError(f"step {i}: the norm weight must be a contiguous float16 tensor of {op.K} elements")
            keep.append(w)
            it.norm_weight, it.norm_eps = w.data_ptr(), float(eps)
    if m is None:
        raise ValueError("a chain starts from a tensor")
    for i, st in enumerate(steps):
        if isinstance(st.output, torch.Tensor):
            o = st.output
            if not o.is_contiguous() or o.device != dev:
                raise ValueError(f"step {i}: output must be a contiguous tensor on the chain's device")
            st.op.check_output(o, m)
            outs[i] = o
            items[i].C = o.data_ptr()
        elif allocate:
            # (every item is a launch that stores its result: `output=False` gets a temporary the caller never sees)
            o = torch.empty((m, st.op.N), dtype=st.op.torch_output_dtype, device=dev)
            if st.output is False:
                keep.append(o)
            else:
                outs[i] = o
            items[i].C = o.data_ptr()
    return items, outs, keep, m, dev


def chain_plan(steps: Sequence[ChainStep], m: Optional[int] = None) -> dict:
    """{"launches": n, "plan": None, "reason": ...}: validates the chain at its row count; the chain runs as its n launches.  Needs
    no device (tensors may live anywhere: only their shapes are read)."""
    items, _, keep, mm, _ = _build(steps, allocate=False)
    for i in range(len(steps)):                          # the plan does not read outputs: any non-NULL stands for "stored"
        if not items[i].C:
            items[i].C = 1
    launches = ctypes.c_int(0)
    plan = _lib.Plan()
    _lib.check(_library().wqaa_chain_plan(items, len(steps), int(m if m is not None else mm), ctypes.byref(launches), ctypes.byref(plan)))
    return {"launches": launches.value, "plan": None, "reason": "a chain runs as the launches it stands for"}


def matmul_chain(steps: Sequence[ChainStep]) -> List[Optional[torch.Tensor]]:
    """Run the chain; returns the steps' outputs (None where `output=False`).  Asynchronous on the current stream."""
    items, outs, keep, m, dev = _build(steps, allocate=True)
    if m == 0:
        return outs
    status = _library().wqaa_matmul_chain(items, len(steps), m, _lib.current_stream_handle(dev))
    if status != _lib.OK:
        _lib.check(status)
    return outs


def _lin_weights(lin):
    cfg = lin.bitblas_matmul.config
    if lin.consistent:
        return (lin.weight, None, None, lin.bias if cfg.with_bias else None)
    return (lin.qweight, lin.scales if cfg.with_scaling else None, lin.zeros if cfg.with_zeros else None,
            lin.bias if cfg.with_bias else None)


class DecoderTail(torch.nn.Module):
    """The post-attention half of a Llama-style decoder layer over four `bitblas_amd.Linear` layers:

        h   = x + o_proj(attn)
        out = h + down_proj(silu(gate_proj(norm(h))) * up_proj(norm(h)))

    (integration/BitNet/modeling_bitnet.py:839-860 with the MLP of :240-244).  At decode row counts: three launches with the
    elementwise ops folded in (`forward_ex`, `matmul_gate_up`); the layers' plain launches with torch's elementwise kernels
    elsewhere.  `persistent` is accepted for callers of rounds 3-5 and ignored (the one-launch member is gone).  The layers keep
    their own buffers and state_dict keys."""

    def __init__(self, o_proj, gate_proj, up_proj, down_proj, norm_weight: torch.Tensor, eps: float = 1e-6, persistent: bool = False):
        super().__init__()
        self.o_proj, self.gate_proj, self.up_proj, self.down_proj = o_proj, gate_proj, up_proj, down_proj
        self.norm_weight = norm_weight
        self.eps = float(eps)
        self.persistent = bool(persistent)

    def steps(self, attn, x, h_out=None, out=None):
        mm = lambda lin: lin.bitblas_matmul  # noqa: E731
        return [
            ChainStep(mm(self.o_proj), _lin_weights(self.o_proj), attn, residual=x, output=h_out if h_out is not None else False),
            ChainStep(mm(self.gate_proj), _lin_weights(self.gate_proj), 0, norm=(self.norm_weight, self.eps), output=False,
                      up_op=mm(self.up_proj), up_weights=_lin_weights(self.up_proj)),
            ChainStep(mm(self.down_proj), _lin_weights(self.down_proj), 1, residual=0, output=out),
        ]

    def forward(self, attn, x):
        from .group import matmul_gate_up
        m = attn.numel() // attn.shape[-1]
        ops = [lin.bitblas_matmul for lin in (self.o_proj, self.gate_proj, self.up_proj, self.down_proj)]
        if m >= 1 and all(op.fused_ops_supported(m) for op in ops):
            a2 = attn.reshape(m, attn.shape[-1])
            x2 = x.reshape(m, x.shape[-1]).contiguous()
            h = self.o_proj.forward_ex(a2, residual=x2)
            act = matmul_gate_up(ops[1], ops[2], h, _lin_weights(self.gate_proj), _lin_weights(self.up_proj), norm=(self.norm_weight, self.eps))
            return self.down_proj.forward_ex(act, residual=h).reshape(x.shape)
        h = x + self.o_proj(attn)
        hn = torch.nn.functional.rms_norm(h, (h.shape[-1],), self.norm_weight, self.eps)
        act = torch.nn.functional.silu(self.gate_proj(hn)) * self.up_proj(hn)
        return h + self.down_proj(act)
