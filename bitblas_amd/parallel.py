"""Column (N) sharding of a quantised matmul across GPUs + the all-gather of output slices.

The reference has no multi-GPU code (no NCCL / torch.distributed call sites).
The path shards naturally: output column n depends only on W[n, :], Scale[n, :], Zeros[n, :] (column
n of QZeros) and Bias[n] (tirscript/matmul_dequantize_impl.py:453-459).  Rank p owns rows
[p*N/P, (p+1)*N/P) and runs the unmodified single-GPU kernel with N' = N/P; the only collective is
one all-gather of the [M, N/P] slices (RCCL over xGMI under backend "nccl"; gloo in the CPU tests).
One process per GPU; A is replicated.
"""
from __future__ import annotations

from dataclasses import replace
from typing import Callable, Optional

import torch
import torch.distributed as dist

from .matmul import Matmul, MatmulConfig, torch_dtype


def shard_bounds(N: int, rank: int, world: int, bits: int = 16, quantized_zeros: bool = False):
    """Row range of rank `rank`.  N/P must stay Linear-legal (multiple of 16, module/__init__.py:155-159)
    and, for quantized zeros, a whole number of packed bytes."""
    if N % world != 0:
        raise ValueError(f"N={N} is not divisible by world size {world}")
    per = N // world
    if per % 16 != 0:
        raise ValueError(f"N/P={per} must be a multiple of 16")
    if quantized_zeros and (per * bits) % 8 != 0:
        raise ValueError(f"N/P={per} x {bits} bit does not fill whole bytes of QZeros")
    return rank * per, (rank + 1) * per


def shard_operands(rank: int, world: int, *, W, bits: int, scale=None, zeros=None, bias=None,
                   zeros_mode: str = "original"):
    """Slice the (already transformed) operands of a full-size matmul for one rank."""
    N = W.shape[0]
    lo, hi = shard_bounds(N, rank, world, bits, zeros is not None and zeros_mode == "quantized")
    out = {"W": W[lo:hi].contiguous()}
    out["scale"] = None if scale is None else scale[lo:hi].contiguous()
    if zeros is None:
        out["zeros"] = None
    elif zeros_mode == "quantized":   # (K/g, N*bits/8): shard the packed second axis
        out["zeros"] = zeros[:, lo * bits // 8: hi * bits // 8].contiguous()
    else:
        out["zeros"] = zeros[lo:hi].contiguous()
    out["bias"] = None if bias is None else bias[lo:hi].contiguous()
    return out


def gather_columns(local_out: torch.Tensor, group=None, out: Optional[torch.Tensor] = None,
                   staging: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[.., N/P] slices -> [.., N] on every rank with ONE all-gather.

    RCCL (like NCCL and gloo) gathers rank-major: the result of a collective is [P, rows, N/P].  For a single row
    (decode, M = 1) or a single rank that already IS the [rows, N] row-major output, so the collective writes straight into
    `out` - no staging, no copy.  For rows > 1 the slices have to be interleaved: one strided copy staging -> out (the only
    copy; callers that stream row blocks - ColumnParallelMatmul - run it on the communication stream under the next
    block's GEMM).  `staging`: optional [P * rows, N/P] scratch to reuse."""
    world = dist.get_world_size(group)
    lead, per = local_out.shape[:-1], local_out.shape[-1]
    flat = local_out.reshape(-1, per)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    rows = flat.shape[0]
    if out is None:
        out = torch.empty(*lead, world * per, dtype=flat.dtype, device=flat.device)
    elif tuple(out.shape) != (*lead, world * per) or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous {(*lead, world * per)} tensor")
    if rows == 1 or world == 1:
        dist.all_gather_into_tensor(out.view(world * rows, per), flat, group=group)
        return out
    if staging is None:
        staging = torch.empty((world * rows, per), dtype=flat.dtype, device=flat.device)
    else:
        staging = staging.view(-1)[: world * rows * per].view(world * rows, per)
    dist.all_gather_into_tensor(staging, flat, group=group)
    out.view(rows, world, per).copy_(staging.view(world, rows, per).permute(1, 0, 2))
    return out


class ColumnParallelMatmul:
    """`Matmul` over an N/P shard + all-gather of the [M, N/P] slices.

    M = 1: one kernel, one small all-gather straight into the output.  Large M (prefill; BASELINE c5: M = 4096 fp8,
    58.7 MB slices at N = 57344): the rows are processed in blocks of `row_block`; the all-gather + interleave of block
    i runs on a communication stream while the compute stream is already in block i + 1's GEMM - xGMI is point-to-point
    (7 links x ~153 GB/s per GPU), so a gather of the whole slice at the end would leave the links idle during the
    GEMM and the matrix cores idle during the gather.
    `compute` lets the CPU tests swap the kernel launch for the oracle; the default is the HIP operator."""

    @staticmethod
    def auto_row_block(M: int, n_local: int, cus: int = 256) -> int:
        """Rows per pipeline block: enough for every CU to own a 256 x 256 tile of the block's GEMM (a block that leaves
        CUs idle costs more than the overlap buys), at least two blocks once M >= 2048 so the first gather starts at half
        of the compute.  Per-rank shards at P = 8 (N/P = 1024 ... 3584) end up with two blocks of 2048 rows - there the
        gather (58.7 MB per rank over 7 xGMI links) outweighs the GEMM anyway."""
        tiles_per_256 = max(1, (n_local + 255) // 256)
        rows = 256 * ((cus + tiles_per_256 - 1) // tiles_per_256)
        rows = max(512, rows)
        if rows >= M:
            rows = M // 2 if M >= 2048 else M
        return int(rows)

    def __init__(self, config: MatmulConfig, group=None, compute: Optional[Callable] = None, row_block: Optional[int] = None,
                 direct_store: bool = False, window_slots: int = 2, window_timeout_ms: int = 2000, **matmul_kwargs):
        """`direct_store=True`: a single row (decode, M = 1) skips the collective - the kernel writes this rank's slice into its
        own PEER WINDOW row, one exchange launch stores it into every peer's row and waits for theirs (bitblas_amd/peer.py;
        hipIpc-mapped device memory on GPUs, a shared mapping in the CPU tests).  The returned `[.., N]` row is a view of the
        window: valid until `window_slots - 1` more single-row calls have been enqueued (pass `out=` to get a copy).  Rows > 1 and
        world = 1 take the all-gather path as before.  Not for hipGraph capture (the step number is a launch argument)."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.full_config = config
        src_bits = Matmul.BITBLAS_TRICK_DTYPE_MAP[config.W_dtype][1]
        self.lo, self.hi = shard_bounds(config.N, self.rank, self.world, src_bits,
                                        config.with_zeros and config.zeros_mode == "quantized")
        self.local_config = replace(config, N=config.N // self.world)
        m_static = config.M if isinstance(config.M, int) else 4096
        self.row_block = int(row_block) if row_block else self.auto_row_block(m_static, config.N // self.world)
        self._compute = compute
        self._matmul_kwargs = matmul_kwargs
        self._ops = {}
        self.op = None if compute is not None else self._op_for(config.M if isinstance(config.M, int) else None)
        self._comm_stream = None
        self._staging = [None, None]
        self.direct_store = bool(direct_store)
        self._window = None
        self._window_args = (int(window_slots), int(window_timeout_ms))

    def _op_for(self, rows):
        """operator for a block of `rows` rows (static-M configs get one operator per block height)"""
        if rows is None or not isinstance(self.local_config.M, int):
            key = None
            cfg = self.local_config
        else:
            key = rows
            cfg = replace(self.local_config, M=rows)
        if key not in self._ops:
            self._ops[key] = Matmul(cfg, enable_tuning=False, **self._matmul_kwargs)
        return self._ops[key]

    def _local(self, A, W, scale, zeros, bias, output=None):
        if self._compute is not None:
            r = self._compute(A, W, scale, zeros, bias)
            if output is not None:
                output.copy_(r)
                return output
            return r
        return self._op_for(A.shape[0] if A.dim() == 2 else None)(A, W, scale=scale, zeros=zeros, bias=bias, output=output)

    def _forward_direct(self, A, W, scale, zeros, bias, out):
        """one row, no collective: GEMV into the own window row, exchange, the window row is the result"""
        from .peer import make_window
        per = self.local_config.N
        dtype = torch_dtype(self.full_config.out_dtype)
        item = torch.empty((), dtype=dtype).element_size()
        if self._window is None:
            self._window = make_window(self.group, self.world * per * item, A.device, *self._window_args)
        win = self._window
        step = win.next_step()
        row = win.row(win.slot_of(step), dtype)                     # [N]
        own = row[self.rank * per:(self.rank + 1) * per].view(1, per)
        self._local(A.reshape(1, A.shape[-1]), W, scale, zeros, bias, output=own)
        win.exchange(step, self.rank * per * item, per * item)
        res = row.view(*A.shape[:-1], self.world * per)
        if out is not None:
            if tuple(out.shape) != tuple(res.shape) or out.dtype != res.dtype:
                raise ValueError(f"out must be a {tuple(res.shape)} {res.dtype} tensor")
            out.copy_(res)
            return out
        return res

    def check_peers(self):
        """raise `peer.PeerTimeout` if an exchange of the direct-store path gave up waiting (synchronises on GPUs)"""
        if self._window is not None:
            self._window.check()

    def close(self) -> None:
        """release the peer-store window (hipIpc mappings of the peers' windows + the own uncached allocation); idempotent"""
        win, self._window = self._window, None
        if win is not None:
            win.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown: the driver releases the mappings with the process
            pass

    def forward(self, A, W, scale=None, zeros=None, bias=None, out=None):
        rows = A.numel() // A.shape[-1]
        if self.direct_store and rows == 1 and self.world > 1:
            return self._forward_direct(A, W, scale, zeros, bias, out)
        blocked = A.dim() == 2 and rows > self.row_block and self.world > 1
        if not blocked:
            return gather_columns(self._local(A, W, scale, zeros, bias), self.group, out=out)
        per = self.local_config.N
        if out is None:
            out = torch.empty((rows, self.world * per), dtype=torch_dtype(self.full_config.out_dtype), device=A.device)
        local = torch.empty((rows, per), dtype=out.dtype, device=A.device)
        on_gpu = A.is_cuda
        if on_gpu:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=A.device)
            comm, cur = self._comm_stream, torch.cuda.current_stream(A.device)
            comm.wait_stream(cur)                  # `out` / staging allocations above are ordered before their first use
        for i, r0 in enumerate(range(0, rows, self.row_block)):
            r1 = min(rows, r0 + self.row_block)
            self._local(A[r0:r1], W, scale, zeros, bias, output=local[r0:r1])
            need = self.world * (r1 - r0) * per
            stg = self._staging[i & 1]
            if stg is None or stg.numel() < need or stg.dtype != out.dtype or stg.device != out.device:
                stg = self._staging[i & 1] = torch.empty(self.world * self.row_block * per, dtype=out.dtype, device=A.device)
            if on_gpu:
                ev = torch.cuda.Event()
                ev.record(cur)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)            # block i's GEMM is done; block i + 1's runs on `cur` meanwhile
                    gather_columns(local[r0:r1], self.group, out=out[r0:r1], staging=stg)
            else:
                gather_columns(local[r0:r1], self.group, out=out[r0:r1], staging=stg)
        if on_gpu:
            cur.wait_stream(comm)
            local.record_stream(comm)
        return out

    __call__ = forward


class ColumnParallelGroup:
    """The projections of a decoder layer that read the same input (q/k/v, gate/up), every one column-sharded over the
    ranks: ONE kernel launch per rank for the whole group (`matmul_group`, include/wqaa.h wqaa_matmul_group) and ONE
    all-gather for all of its outputs.

    At M = 1 the collective, not the kernel, is the cost of a column-parallel layer: a 2-16 KB all-gather is a fixed
    RCCL latency of tens of microseconds against ~7 us for the grouped GEMV, and xGMI is point-to-point - three small
    gathers queue behind each other.  Every rank writes its slices of all members side by side into one staging row
    block `[rows, sum_i N_i / P]` (the group launch gives each member its own output pointer inside it), the staging is
    gathered once, and member i's `[rows, N_i]` output is the rank-major view `gathered[:, :, off_i : off_i + N_i / P]`
    - returned as a strided view (`as_views=True`: no copy; rank p's columns are contiguous, which is what per-head
    consumers read) or copied out contiguous (default).  SURVEY.md section 8(e); the reference has no multi-GPU code.
    `compute` lets the CPU tests swap the kernel launch for the oracle."""

    def __init__(self, configs, group=None, compute: Optional[Callable] = None, **matmul_kwargs):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.full_configs = list(configs)
        if not self.full_configs:
            raise ValueError("a group needs at least one member")
        out_dtypes = {c.out_dtype for c in self.full_configs}
        if len(out_dtypes) != 1:
            raise ValueError("the members of a group share one out_dtype (one staging buffer is gathered)")
        self.bounds = []
        for c in self.full_configs:
            bits = Matmul.BITBLAS_TRICK_DTYPE_MAP[c.W_dtype][1]
            self.bounds.append(shard_bounds(c.N, self.rank, self.world, bits, c.with_zeros and c.zeros_mode == "quantized"))
        self.local_configs = [replace(c, N=c.N // self.world) for c in self.full_configs]
        self.pers = [c.N for c in self.local_configs]
        self.offs = [sum(self.pers[:i]) for i in range(len(self.pers))]
        self.per_total = sum(self.pers)
        self._compute = compute
        self.ops = None if compute is not None else [Matmul(c, enable_tuning=False, **matmul_kwargs) for c in self.local_configs]
        self._out_dtype = torch_dtype(self.full_configs[0].out_dtype)

    def forward(self, A, weights, as_views: bool = False):
        """`weights[i]`: the rank's shard of member i as `Matmul.forward` takes it after A - `W` or `(W, scale, zeros, bias)`
        (see `shard_operands`).  Returns the members' full `[..., N_i]` outputs, identical on every rank."""
        lead = A.shape[:-1]
        rows = A.numel() // A.shape[-1]
        world, P = self.world, self.per_total
        ws = [((w,) if isinstance(w, torch.Tensor) else tuple(w)) for w in weights]
        ws = [(w + (None,) * 4)[:4] for w in ws]
        # every member's slice goes into its own contiguous block of ONE flat staging buffer: block i is [rows, N_i / P]
        staging = torch.empty(rows * P, dtype=self._out_dtype, device=A.device)
        blocks, off = [], 0
        for n in self.pers:
            blocks.append(staging[off:off + rows * n].view(rows, n))
            off += rows * n
        A2 = A.reshape(rows, A.shape[-1])
        if self._compute is not None:
            for b, w in zip(blocks, ws):
                b.copy_(self._compute(A2, *w))
        else:
            from .group import matmul_group
            matmul_group(self.ops, A2, ws, outputs=blocks)
        gathered = torch.empty(world * rows * P, dtype=self._out_dtype, device=A.device)
        dist.all_gather_into_tensor(gathered, staging, group=self.group)           # the ONE collective of the group
        g = gathered.view(world, rows * P)
        outs, off = [], 0
        for n in self.pers:
            v = g[:, off:off + rows * n].view(world, rows, n).permute(1, 0, 2)     # [rows, P, N_i / P]: rank-major columns
            off += rows * n
            outs.append(v if as_views else v.reshape(*lead, world * n))
        return outs

    __call__ = forward


class ColumnParallelLinear(torch.nn.Module):
    """`bitblas.Linear` column-sharded over the ranks: rank p owns output features `[p N/P, (p+1) N/P)` as an ordinary
    `Linear(in_features, out_features / P)` (same buffers, same kernels) and `forward` all-gathers the `[..., N/P]` slices
    (`gather_output=False` returns the local slice, for a row-parallel consumer).

    `load_full_state_dict` takes the state_dict of the UNSHARDED layer - the reference's checkpoint layout: `qweight`
    `(N, K*bits/8)`, `scales` / `zeros` `(N, K/g)` or packed `zeros` `(K/g, N*bits/8)`, `bias` `(N,)`
    (bitblas/module/__init__.py:164-205) - and keeps this rank's rows of each: a checkpoint written by upstream BitBLAS
    (or by `Linear.repack_from_gptq`) loads on any world size without re-quantisation.  The reference has no multi-GPU
    code; sharding rule = `shard_operands`."""

    def __init__(self, in_features: int, out_features: int, group=None, gather_output: bool = True, **linear_kwargs):
        super().__init__()
        from .module import Linear
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.in_features, self.out_features = in_features, out_features
        self.gather_output = gather_output
        w_dtype = linear_kwargs.get("W_dtype", "float16")
        self.bits = Matmul.BITBLAS_TRICK_DTYPE_MAP[w_dtype][1]
        self.quantized_zeros = bool(linear_kwargs.get("with_zeros")) and linear_kwargs.get("zeros_mode") == "quantized"
        self.lo, self.hi = shard_bounds(out_features, self.rank, self.world, self.bits, self.quantized_zeros)
        self.local = Linear(in_features, out_features // self.world, **linear_kwargs)

    def shard_state_dict(self, full: dict) -> dict:
        """this rank's slice of an unsharded `Linear.state_dict()` (keys without a prefix)"""
        lo, hi, bits = self.lo, self.hi, self.bits
        out = {}
        for key, t in full.items():
            if key == "zeros" and self.quantized_zeros:
                if t.shape[-1] * 8 != self.out_features * bits:
                    raise ValueError(f"zeros: packed width {t.shape[-1]} does not hold {self.out_features} {bits}-bit zero points")
                out[key] = t[:, lo * bits // 8: hi * bits // 8].contiguous()
            elif key in ("qweight", "weight", "scales", "zeros", "bias"):
                if t.shape[0] != self.out_features:
                    raise ValueError(f"{key}: {t.shape[0]} rows, the unsharded layer has {self.out_features}")
                out[key] = t[lo:hi].contiguous()
            else:
                raise KeyError(f"unexpected key {key!r} in a Linear state_dict")
        return out

    def load_full_state_dict(self, full: dict, strict: bool = True):
        return self.local.load_state_dict(self.shard_state_dict(full), strict=strict)

    def forward(self, A, out: Optional[torch.Tensor] = None):
        local = self.local(A)
        if not self.gather_output:
            return local
        return gather_columns(local, self.group, out=out)
