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
                 **matmul_kwargs):
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

    def forward(self, A, W, scale=None, zeros=None, bias=None, out=None):
        rows = A.numel() // A.shape[-1]
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
