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

from .matmul import Matmul, MatmulConfig


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


def gather_columns(local_out: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[.., N/P] slices -> [.., N] on every rank: one all-gather (+ a strided copy when M > 1)."""
    world = dist.get_world_size(group)
    lead, per = local_out.shape[:-1], local_out.shape[-1]
    flat = local_out.reshape(-1, per).contiguous()
    # concatenated along dim 0 (the form both RCCL and gloo accept), viewed as [P, rows, N/P]
    staged = torch.empty((world * flat.shape[0], per), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(staged, flat, group=group)
    full = staged.view(world, flat.shape[0], per).permute(1, 0, 2).reshape(flat.shape[0], world * per)
    full = full.reshape(*lead, world * per)
    if out is not None:
        out.copy_(full)
        return out
    return full.contiguous()


class ColumnParallelMatmul:
    """`Matmul` over an N/P shard + all-gather.  `compute` lets the CPU tests swap the kernel launch
    for the oracle; the default is the HIP operator."""

    def __init__(self, config: MatmulConfig, group=None, compute: Optional[Callable] = None, **matmul_kwargs):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.full_config = config
        src_bits = Matmul.BITBLAS_TRICK_DTYPE_MAP[config.W_dtype][1]
        self.lo, self.hi = shard_bounds(config.N, self.rank, self.world, src_bits,
                                        config.with_zeros and config.zeros_mode == "quantized")
        self.local_config = replace(config, N=config.N // self.world)
        self._compute = compute
        self.op = None if compute is not None else Matmul(self.local_config, enable_tuning=False, **matmul_kwargs)

    def forward(self, A, W, scale=None, zeros=None, bias=None):
        if self._compute is not None:
            local = self._compute(A, W, scale, zeros, bias)
        else:
            local = self.op(A, W, scale=scale, zeros=zeros, bias=bias)
        return gather_columns(local, self.group)

    __call__ = forward
