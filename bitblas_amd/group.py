"""Groups of independent operators in one launch: `matmul_group`, `LinearGroup`.

The reference runs the projections of a decoder layer that share an input as separate operator calls, or fuses them by
concatenating their weights along N before quantisation (integration/BitNet/modeling_bitnet.py:
`BitnetAttentionQKVFused.from_bit_attention` :496-517, `BitnetMLPFuseGateUp.from_bit_mlp` :271-280, on by default in
`quantize(fuse_qkv=True, fuse_gateup=True)` :1433-1445).  On MI355X a dependent kernel boundary costs ~1.3 us against
~4 us for a whole 4096 x 4096 int4 GEMV, so at decode batch sizes the boundaries are a third of a layer.
`wqaa_matmul_group` (include/wqaa.h) is the launch-level form of the same fusion: the members keep their own packed
tensors (no re-quantisation, no concatenated copy, q/k/v of different widths are fine) and run as ONE kernel launch
whenever they share a GEMV tile configuration (same K / dtypes / format / group size / flags, M <= 2); anything else
runs member by member in order, so the call is always equivalent to calling each operator in turn - bit for bit
(tests/test_group_gpu.py).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Union

import torch

from . import lib as _lib
from .matmul import Matmul, check_norm, rms_norm_reference

GROUP_MAX = 8


class GroupItem(ctypes.Structure):
    """struct wqaa_group_item (include/wqaa.h)."""
    _fields_ = [("desc", ctypes.POINTER(_lib.MatmulDesc)), ("A", ctypes.c_void_p), ("B", ctypes.c_void_p),
                ("LUT", ctypes.c_void_p), ("Scale", ctypes.c_void_p), ("Zeros", ctypes.c_void_p),
                ("Bias", ctypes.c_void_p), ("C", ctypes.c_void_p)]


_bound = False


def _library():
    global _bound
    lib = _lib.load_library()
    if not _bound:
        lib.wqaa_matmul_group.restype = ctypes.c_int
        lib.wqaa_matmul_group.argtypes = [ctypes.POINTER(GroupItem), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.wqaa_matmul_group_ex.restype = ctypes.c_int
        lib.wqaa_matmul_group_ex.argtypes = [ctypes.POINTER(GroupItem), ctypes.POINTER(ctypes.POINTER(_lib.Epilogue)), ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p]
        lib.wqaa_matmul_gate_up.restype = ctypes.c_int
        lib.wqaa_matmul_gate_up.argtypes = [ctypes.POINTER(GroupItem), ctypes.POINTER(GroupItem), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.POINTER(_lib.Epilogue)]
        lib.wqaa_gate_up_plan.restype = ctypes.c_int
        lib.wqaa_gate_up_plan.argtypes = [ctypes.POINTER(_lib.MatmulDesc), ctypes.c_int, ctypes.c_int, ctypes.POINTER(_lib.Plan)]
        lib.wqaa_group_plan.restype = ctypes.c_int
        lib.wqaa_group_plan.argtypes = [ctypes.POINTER(ctypes.POINTER(_lib.MatmulDesc)), ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_lib.Plan)]
        _bound = True
    return lib


def group_plan(ops: Sequence[Matmul], m: int = 1) -> dict:
    """How a group of operators would run at `m` rows: {"launches": 1 (fused) | len(ops), "plan": {...} | None}.
    Needs no device."""
    lib = _library()
    n = len(ops)
    descs = (ctypes.POINTER(_lib.MatmulDesc) * n)(*[ctypes.pointer(op.lib.desc) for op in ops])
    launches = ctypes.c_int(0)
    plan = _lib.Plan()
    _lib.check(lib.wqaa_group_plan(descs, n, int(m), ctypes.byref(launches), ctypes.byref(plan)))
    return {"launches": launches.value, "plan": plan.as_dict() if launches.value == 1 and n > 1 else None}


def _log_group(descs, m):
    """WQAA_PLAN_LOG (lib._log_plan): a fused group logs its group plan once, an unfused one its members"""
    key = (tuple(bytes(d) for d in descs), int(m))
    if key in _lib._plan_logged:
        return
    _lib._plan_logged.add(key)
    lib = _library()
    n = len(descs)
    arr = (ctypes.POINTER(_lib.MatmulDesc) * n)(*[ctypes.pointer(d) for d in descs])
    launches = ctypes.c_int(0)
    plan = _lib.Plan()
    if lib.wqaa_group_plan(arr, n, int(m), ctypes.byref(launches), ctypes.byref(plan)) == _lib.OK and launches.value == 1 and n > 1:
        with open(_lib._PLAN_LOG, "a") as f:
            f.write(f"{int(m)}\t{plan.as_dict()['name']}\n")
    else:
        for d in descs:
            _lib._log_plan(d, m)


def _as_list(x, n, what):
    if isinstance(x, torch.Tensor) or x is None:
        return [x] * n
    x = list(x)
    if len(x) != n:
        raise ValueError(f"{what}: expected {n} entries, got {len(x)}")
    return x


def matmul_group(ops: Sequence[Matmul], A: Union[torch.Tensor, Sequence[torch.Tensor]], weights: Sequence,
                 outputs: Optional[Sequence[Optional[torch.Tensor]]] = None, norm=None) -> List[torch.Tensor]:
    """`[op(A_i, *weights_i) for op in ops]` with as few launches as the library can do it in.

    ops      the operators (every member must have been built for the row count of its A);
    A        one tensor shared by all members, or one per member (all with the same row count);
    weights  per member the arguments of `Matmul.forward` after A: `W` or `(W, scale, zeros, bias)` (trailing ones optional);
    outputs  optional preallocated outputs (contiguous, on A's device); they must not overlap;
    norm     (weight, eps): A (ONE tensor) is the hidden state in front of the layer's RMSNorm, every member computes
             `op(rms_norm(A), ...)` (include/wqaa.h WQAA_EPI_RMSNORM_INPUT) - inside the group's launch where
             `Matmul.norm_supported`, by torch's kernels in front of it elsewhere.
    Asynchronous on the current stream of A's device, like `Matmul.forward`."""
    n = len(ops)
    if n == 0:
        return []
    if norm is not None:
        if not isinstance(A, torch.Tensor):
            raise ValueError("a norm in front of a group reads ONE hidden state")
        m_rows = ops[0].check_activation(A)
        check_norm(norm, A, ops[0].K)
        # (members the library cannot fuse into one launch still run one by one, each with the norm in its own staging pass)
        if not all(op.norm_supported(m_rows) and op.K == ops[0].K for op in ops) or m_rows == 0:
            A, norm = rms_norm_reference(A, *norm), None
    if len(weights) != n:
        raise ValueError(f"weights: expected {n} entries, got {len(weights)}")
    As = _as_list(A, n, "A")
    outs = _as_list(outputs, n, "outputs") if outputs is not None else [None] * n
    items = (GroupItem * n)()
    m = None
    keep = []
    dev = As[0].device
    for i, (op, a, w) in enumerate(zip(ops, As, weights)):
        mi = op.check_activation(a)
        if m is None:
            m = mi
        elif mi != m:
            raise ValueError(f"member {i} has {mi} activation rows, member 0 has {m}: a group shares one row count")
        if a.device != dev:
            raise ValueError("all members of a group run on one device")
        if not a.is_contiguous():
            a = a.contiguous()
        w = (w,) if isinstance(w, torch.Tensor) else tuple(w)
        W, scale, zeros, bias = (w + (None,) * 4)[:4]
        if W.numel() * W.element_size() != op._w_bytes:
            raise ValueError(f"member {i}: W holds {W.numel() * W.element_size()} bytes, the operator expects {op._w_bytes} "
                             f"(shape {op.retrieve_weight_shape()}: run transform_weight first)")
        out = outs[i]
        if out is None:
            out = torch.empty(a.shape[:-1] + (op.N,), dtype=op.torch_output_dtype, device=a.device)
        elif not out.is_contiguous() or out.device != a.device:
            raise ValueError("outputs must be contiguous tensors on A's device")
        else:
            op.check_output(out, mi)
        outs[i] = out
        lut = op._ensure_lut(a.device)
        keep.append((a, lut))
        it = items[i]
        it.desc = ctypes.pointer(op.lib.desc)
        it.A, it.B, it.C = a.data_ptr(), W.data_ptr(), out.data_ptr()
        it.LUT = lut.data_ptr() if lut is not None else None
        it.Scale = scale.data_ptr() if scale is not None else None
        it.Zeros = zeros.data_ptr() if zeros is not None else None
        it.Bias = bias.data_ptr() if bias is not None else None
    if m == 0:
        return outs
    stream = _lib.current_stream_handle(dev)
    # members that need scratch (split-K MFMA members, m > 2) are never fused: run them through their own fast path,
    # which hands the library a caller-owned workspace
    if any(op.lib.workspace_need(m) for op in ops):
        for i, op in enumerate(ops):
            it = items[i]
            op.lib.run(it.A, it.B, it.LUT, it.Scale, it.Zeros, it.Bias, it.C, m, stream, dev)
        return outs
    if _lib._PLAN_LOG:
        _log_group([op.lib.desc for op in ops], m)
    if norm is not None:
        epi = _lib.norm_epilogue(norm[0].data_ptr(), norm[1])
        epis = (ctypes.POINTER(_lib.Epilogue) * n)(*[ctypes.pointer(epi)] * n)
        status = _library().wqaa_matmul_group_ex(items, epis, n, m, stream)
        if status == _lib.ERR_UNSUPPORTED:        # the selector's word is final: torch's norm in front of the plain group
            normed = rms_norm_reference(As[0], *norm)
            return matmul_group(ops, normed, weights, outputs=outs)
    else:
        status = _library().wqaa_matmul_group(items, n, m, stream)
    if status != _lib.OK:
        _lib.check(status)
    return outs


def gate_up_plan(op: Matmul, m: int = 1, norm: bool = False):
    """plan of the one-launch `matmul_gate_up` of two operators like `op` at `m` rows (`norm`: with the RMSNorm in front), or
    None where it does not exist (then the group launch + torch's `silu`, `mul` run).  Needs no device."""
    plan = _lib.Plan()
    if _library().wqaa_gate_up_plan(ctypes.byref(op.lib.desc), int(m), 1 if norm else 0, ctypes.byref(plan)) != _lib.OK:
        return None
    return plan.as_dict()


def _pair_planned(op: Matmul, m: int, norm: bool) -> bool:
    """`gate_up_plan(...) is not None`, asked once per (operator, m, norm): the forward path of a decode step made one or two ctypes
    plan calls per call.  The verdict is a plan-time one (the tuning variables are read when an operator is built)."""
    cache = op.__dict__.setdefault("_pair_planned", {})
    key = (int(m), bool(norm))
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = gate_up_plan(op, m, norm=norm) is not None
    return hit


def matmul_gate_up(gate_op: Matmul, up_op: Matmul, A: torch.Tensor, gate_weights, up_weights,
                   output: Optional[torch.Tensor] = None, norm=None) -> torch.Tensor:
    """`F.silu(gate_op(A, *gate_weights)) * up_op(A, *up_weights)` - the gated activation of a Llama-style MLP (the reference's
    callers: integration/BitNet/modeling_bitnet.py:240-244, :281-287) with both projections in ONE launch that stores only the
    activation (wqaa_matmul_gate_up: float16 x 1 / 2 / 4-bit integer weights at m <= 2); elsewhere the group launch followed
    by torch's two elementwise kernels.  weights: `W` or `(W, scale, zeros, bias)` per projection, as `matmul_group`.
    norm = (weight, eps): A is the hidden state in front of the MLP's RMSNorm (`post_attention_layernorm`, :858), folded in too."""
    m = gate_op.check_activation(A)
    if up_op.check_activation(A) != m or bytes(gate_op.lib.desc) != bytes(up_op.lib.desc):
        raise ValueError("gate and up must be operators of one configuration (same N, K, formats, group size)")
    if norm is not None:
        check_norm(norm, A, gate_op.K)
        if not gate_op.fused_ops_supported(m) or not _pair_planned(gate_op, m, True):     # the selector's word
            A, norm = rms_norm_reference(A, *norm), None
    # the selector's word is final: where it refuses the pair (no member, the LDS limit) the group launch + torch's two kernels run
    if not gate_op.fused_ops_supported(m) or m == 0 or not _pair_planned(gate_op, m, norm is not None):
        if norm is not None:
            A, norm = rms_norm_reference(A, *norm), None
        g, u = matmul_group([gate_op, up_op], A, [gate_weights, up_weights])
        return torch.mul(torch.nn.functional.silu(g), u, out=output)
    if output is None:
        output = torch.empty(A.shape[:-1] + (gate_op.N,), dtype=gate_op.torch_output_dtype, device=A.device)
    elif not output.is_contiguous() or output.device != A.device:
        raise ValueError("output must be a contiguous tensor on A's device")
    else:
        gate_op.check_output(output, m)
    if not A.is_contiguous():
        A = A.contiguous()
    items = (GroupItem * 2)()
    for it, op, w in ((items[0], gate_op, gate_weights), (items[1], up_op, up_weights)):
        w = (w,) if isinstance(w, torch.Tensor) else tuple(w)
        W, scale, zeros, bias = (w + (None,) * 4)[:4]
        if W.numel() * W.element_size() != op._w_bytes:
            raise ValueError(f"W holds {W.numel() * W.element_size()} bytes, the operator expects {op._w_bytes} "
                             f"(shape {op.retrieve_weight_shape()}: run transform_weight first)")
        it.desc = ctypes.pointer(op.lib.desc)
        it.A, it.B, it.C = A.data_ptr(), W.data_ptr(), None
        it.Scale = scale.data_ptr() if scale is not None else None
        it.Zeros = zeros.data_ptr() if zeros is not None else None
        it.Bias = bias.data_ptr() if bias is not None else None
    if _lib._PLAN_LOG:
        key = (bytes(gate_op.lib.desc), int(m), "pair", norm is not None)
        if key not in _lib._plan_logged:
            _lib._plan_logged.add(key)
            plan = gate_up_plan(gate_op, m, norm=norm is not None)
            if plan is not None:
                with open(_lib._PLAN_LOG, "a") as f:
                    f.write(f"{int(m)}\t{plan['name']}\n")
    epi = _lib.norm_epilogue(norm[0].data_ptr(), norm[1]) if norm is not None else None
    status = _library().wqaa_matmul_gate_up(ctypes.byref(items[0]), ctypes.byref(items[1]), output.data_ptr(), m,
                                            _lib.current_stream_handle(A.device), ctypes.byref(epi) if epi is not None else None)
    if status != _lib.OK:
        _lib.check(status)
    return output


class GatedMLP(torch.nn.Module):
    """`down(act_fn(gate(x)) * up(x))` (+ residual) over three `bitblas_amd.Linear` layers, act_fn = silu - the MLP of a
    Llama-style decoder layer (the reference's callers: integration/BitNet/modeling_bitnet.py:209-244 without its extra
    norm) in TWO launches at decode row counts: `matmul_gate_up` (gate, up and the activation) and `Linear.forward_ex` (down
    and the residual add).  The layers keep their own buffers and state_dict keys; at other row counts / formats the same ops
    run as the layers' plain launches with torch's elementwise kernels between them."""

    def __init__(self, gate_proj, up_proj, down_proj):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = gate_proj, up_proj, down_proj

    @staticmethod
    def _weights(lin):
        cfg = lin.bitblas_matmul.config
        if lin.consistent:
            return (lin.weight, None, None, lin.bias if cfg.with_bias else None)
        return (lin.qweight, lin.scales if cfg.with_scaling else None, lin.zeros if cfg.with_zeros else None,
                lin.bias if cfg.with_bias else None)

    def forward(self, x, residual=None, norm=None):
        """norm = (weight, eps): x is the hidden state in front of the MLP's RMSNorm (`post_attention_layernorm`)"""
        act = matmul_gate_up(self.gate_proj.bitblas_matmul, self.up_proj.bitblas_matmul, x, self._weights(self.gate_proj),
                             self._weights(self.up_proj), norm=norm)
        return self.down_proj.forward_ex(act, residual=residual)


class LinearGroup(torch.nn.Module):
    """`bitblas_amd.Linear` layers that read the same input (q/k/v, gate/up), called as one: `q, k, v = group(x)`.

    The launch-level counterpart of the reference's `fuse_qkv` / `fuse_gateup` (modeling_bitnet.py:1433-1445), which
    concatenates the weights into one `BitLinear` instead; here the layers keep their own buffers and state_dict keys.
    The descriptor / weight-pointer array is built once and only the activation and output pointers change per call."""

    def __init__(self, layers: Sequence[torch.nn.Module]):
        super().__init__()
        if not 1 <= len(layers) <= GROUP_MAX:
            raise ValueError(f"a group holds 1..{GROUP_MAX} layers")
        self.layers = torch.nn.ModuleList(layers)
        self._items = None
        self._keys = None

    def _build(self):
        n = len(self.layers)
        items = (GroupItem * n)()
        keys = []
        for i, layer in enumerate(self.layers):
            if not layer._params_current():
                layer.init_params()
            B, scale, zeros, bias = layer._q_run
            it = items[i]
            it.desc = ctypes.pointer(layer.bitblas_matmul.lib.desc)
            it.B, it.Scale, it.Zeros, it.Bias = B, scale, zeros, bias
            keys.append(layer._q_param_keys)
        self._items, self._keys = items, keys
        mm0 = self.layers[0].bitblas_matmul
        # the layers accept the same activations iff they agree on K, A_dtype and the M they were built for
        self._same_input = all((l.bitblas_matmul._a_cols, l.bitblas_matmul._a_torch_dtype, l.bitblas_matmul.dynamic_range is None,
                                l.bitblas_matmul.config.M) == (mm0._a_cols, mm0._a_torch_dtype, mm0.dynamic_range is None, mm0.config.M)
                               for l in self.layers)
        self._ns = [l.out_features for l in self.layers]
        self._out_dtype = mm0.torch_output_dtype if len({l.bitblas_matmul.torch_output_dtype for l in self.layers}) == 1 else None
        self._nf = [l for l in self.layers if l.source_format == "nf"]

    def forward(self, A, outputs: Optional[Sequence[torch.Tensor]] = None):
        n = len(self.layers)
        mm0 = self.layers[0].bitblas_matmul
        A = mm0.transform_input(A)
        m = mm0.check_activation(A)
        if self._items is None or any(not l._params_current() or l._q_param_keys != k for l, k in zip(self.layers, self._keys)):
            self._build()
        if not self._same_input:
            for layer in self.layers[1:]:
                if layer.bitblas_matmul.check_activation(A) != m:
                    raise ValueError("the layers of a group take the same input")
        if not A.is_contiguous():
            A = A.contiguous()
        items = self._items
        a_ptr = A.data_ptr()
        if outputs is None:
            if self._out_dtype is not None:
                # ONE allocation for the group: the members' (m, N_i) outputs are consecutive contiguous blocks of it
                flat = torch.empty(m * sum(self._ns), dtype=self._out_dtype, device=A.device)
                lead = A.shape[:-1]
                outs = [p.view(lead + (N,)) for p, N in zip(flat.split([m * N for N in self._ns]), self._ns)]
                base, esz = flat.data_ptr(), flat.element_size()
                off = 0
                for i, N in enumerate(self._ns):
                    items[i].A, items[i].C = a_ptr, base + off * esz
                    off += m * N
            else:
                outs = [torch.empty(A.shape[:-1] + (l.out_features,), dtype=l.bitblas_matmul.torch_output_dtype, device=A.device)
                        for l in self.layers]
                for i, o in enumerate(outs):
                    items[i].A, items[i].C = a_ptr, o.data_ptr()
        else:
            outs = list(outputs)
            for i, o in enumerate(outs):
                if not o.is_contiguous() or o.device != A.device:
                    raise ValueError("outputs must be contiguous tensors on A's device")
                self.layers[i].bitblas_matmul.check_output(o, m)
                items[i].A, items[i].C = a_ptr, o.data_ptr()
        for layer in self._nf:
            lut = layer.bitblas_matmul._ensure_lut(A.device)
            items[list(self.layers).index(layer)].LUT = lut.data_ptr()
        if m == 0:
            return tuple(outs)
        stream = _lib.current_stream_handle(A.device)
        if any(l.bitblas_matmul.lib.workspace_need(m) for l in self.layers):
            for i, layer in enumerate(self.layers):
                it = items[i]
                layer.bitblas_matmul.lib.run(it.A, it.B, it.LUT, it.Scale, it.Zeros, it.Bias, it.C, m, stream, A.device)
            return tuple(outs)
        if _lib._PLAN_LOG:
            _log_group([l.bitblas_matmul.lib.desc for l in self.layers], m)
        status = _library().wqaa_matmul_group(items, n, m, stream)
        if status != _lib.OK:
            _lib.check(status)
        return tuple(outs)


__all__ = ["matmul_group", "group_plan", "LinearGroup", "GroupItem", "GROUP_MAX"]
