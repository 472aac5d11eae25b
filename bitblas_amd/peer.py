"""Peer windows: the M = 1 output exchange of a column-parallel matmul without a collective.

Every rank owns a WINDOW - `world` 32-bit flag words, a status word, then `slots` output rows of `row_bytes` - that its peers
can write: device memory mapped through hipIpc handles on GPUs (`HipPeerWindow`: include/wqaa.h wqaa_peer_*, csrc/wqaa_peer.hip),
a shared file mapping on the CPU (`ShmPeerWindow`: what the world-size-2 gloo tests drive - the bookkeeping is the same code).
After the GEMV that wrote this rank's `[1, N/P]` slice into its own row, `exchange` stores the slice into every peer's row at this
rank's columns, posts the step number and waits for the peers' posts of the same step: one launch instead of a small-message
all-gather (tens of microseconds of RCCL latency against ~7 us of kernel).

Bookkeeping (`PeerWindow`): steps count 1, 2, ...; step s uses row slot (s - 1) % slots.  A peer can run at most one step ahead
of this rank's posts (its exchange of step s waits for this rank's post of step s), so while this rank still reads its row of
step s the peer stores into slot s + 1 at most: with two slots the row of step s stays intact until THIS rank posts step s + 1,
i.e. everything enqueued on the stream before the next single-row call reads it safely (`slots` = k: k - 1 more calls).
A late peer ends the wait after `timeout_ms` and is reported by `check()`.

The reference has no multi-GPU path; the sharding is bitblas_amd/parallel.py's."""
from __future__ import annotations

import ctypes
import os
import tempfile
import time
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

FLAG_REGION = 256          # bytes in front of the rows: world flag words, the status word at STATUS_OFF
STATUS_OFF = 128
ROW_ALIGN = 256


class PeerTimeout(RuntimeError):
    pass


class PeerWindow:
    """layout + step / slot bookkeeping shared by the transports"""

    def __init__(self, group, row_bytes: int, slots: int = 2, timeout_ms: int = 2000):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world * 4 > STATUS_OFF:
            raise ValueError(f"world size {self.world} exceeds the flag region ({STATUS_OFF // 4} ranks)")
        if slots < 2:
            raise ValueError("at least two row slots (a peer may still read the previous row)")
        if row_bytes <= 0 or row_bytes % 16:
            raise ValueError("row_bytes: a positive multiple of 16")
        self.row_bytes = int(row_bytes)
        self.row_pitch = (self.row_bytes + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN
        self.slots = int(slots)
        self.timeout_ms = int(timeout_ms)
        self.window_bytes = FLAG_REGION + self.slots * self.row_pitch
        self.step = 0                        # the last step enqueued

    # -- bookkeeping ----------------------------------------------------------------------------------------------------------
    def next_step(self) -> int:
        self.step += 1
        return self.step

    def slot_of(self, step: int) -> int:
        return (step - 1) % self.slots

    def row_offset(self, slot: int) -> int:
        if not 0 <= slot < self.slots:
            raise IndexError(slot)
        return FLAG_REGION + slot * self.row_pitch

    def _check_slice(self, lo: int, nbytes: int):
        if lo < 0 or nbytes <= 0 or lo + nbytes > self.row_bytes or lo % 16 or nbytes % 16:
            raise ValueError(f"slice [{lo}, {lo + nbytes}) of a {self.row_bytes}-byte row must be made of whole, aligned 16-byte pieces")

    # -- transports implement ----------------------------------------------------------------------------------------------------
    def row(self, slot: int, dtype: torch.dtype) -> torch.Tensor:           # this rank's row `slot` as a 1-D tensor
        raise NotImplementedError

    def exchange(self, step: int, lo: int, nbytes: int) -> None:            # push [lo, lo + nbytes) of row slot_of(step), post, wait
        raise NotImplementedError

    def check(self) -> None:                                                # raise PeerTimeout if an exchange gave up
        raise NotImplementedError

    def close(self) -> None:
        pass


class ShmPeerWindow(PeerWindow):
    """CPU transport: every rank's window is a file mapping its peers open (the paths travel by all_gather_object)"""

    def __init__(self, group, row_bytes: int, slots: int = 2, timeout_ms: int = 2000, directory: Optional[str] = None):
        super().__init__(group, row_bytes, slots, timeout_ms)
        base = directory or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
        fd, self.path = tempfile.mkstemp(prefix=f"wqaa_peer_r{self.rank}_", dir=base)
        os.ftruncate(fd, self.window_bytes)
        os.close(fd)
        paths: List[Optional[str]] = [None] * self.world
        dist.all_gather_object(paths, self.path, group=group)
        self.maps = [np.memmap(p, dtype=np.uint8, mode="r+", shape=(self.window_bytes,)) for p in paths]
        self.own = self.maps[self.rank]
        self.flags = self.own[: self.world * 4].view(np.uint32)
        self.late = 0
        dist.barrier(group=group)            # everybody has opened everybody's file: the names can go
        os.unlink(self.path)

    def row(self, slot: int, dtype: torch.dtype) -> torch.Tensor:
        off = self.row_offset(slot)
        return torch.from_numpy(self.own[off: off + self.row_bytes]).view(dtype)

    def exchange(self, step: int, lo: int, nbytes: int) -> None:
        self._check_slice(lo, nbytes)
        off = self.row_offset(self.slot_of(step)) + lo
        src = self.own[off: off + nbytes]
        for p in range(self.world):
            if p == self.rank:
                continue
            self.maps[p][off: off + nbytes] = src                             # the slice, then the post
            self.maps[p][: self.world * 4].view(np.uint32)[self.rank] = step & 0xFFFFFFFF
        deadline = time.monotonic() + self.timeout_ms / 1000.0
        for p in range(self.world):
            if p == self.rank:
                continue
            while ((int(self.flags[p]) - step) & 0xFFFFFFFF) >= 0x80000000:      # (wrap-around compare, as the kernel's)
                if time.monotonic() > deadline:
                    self.late = 1 + p
                    return
                time.sleep(0)

    def check(self) -> None:
        if self.late:
            raise PeerTimeout(f"rank {self.rank}: the post of rank {self.late - 1} did not arrive within {self.timeout_ms} ms")

    def close(self) -> None:
        self.maps = []
        self.own = None


class _DevMem:
    """a raw device range as something torch can wrap (`__cuda_array_interface__`)"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class HipPeerWindow(PeerWindow):
    """GPU transport: uncached device memory, exported / opened with hipIpc handles, one exchange launch per step"""

    def __init__(self, group, row_bytes: int, device, slots: int = 2, timeout_ms: int = 2000):
        super().__init__(group, row_bytes, slots, timeout_ms)
        from . import lib as L
        self._L = L
        self._lib = L.load_library()
        if self.world > L.PEER_MAX:
            raise ValueError(f"world size {self.world} > {L.PEER_MAX}")
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            base = ctypes.c_void_p()
            L.check(self._lib.wqaa_peer_alloc(self.window_bytes, ctypes.byref(base)))
            self.base = base.value
            handle = ctypes.create_string_buffer(L.PEER_HANDLE_BYTES)
            L.check(self._lib.wqaa_peer_export(self.base, handle))
            handles: List[Optional[bytes]] = [None] * self.world
            dist.all_gather_object(handles, handle.raw, group=group)
            self.peer_base: List[Optional[int]] = [None] * self.world
            for p, h in enumerate(handles):
                if p == self.rank:
                    self.peer_base[p] = self.base
                    continue
                ptr = ctypes.c_void_p()
                L.check(self._lib.wqaa_peer_open(ctypes.create_string_buffer(h, L.PEER_HANDLE_BYTES), ctypes.byref(ptr)))
                self.peer_base[p] = ptr.value
        self._mem = torch.as_tensor(_DevMem(self.base, self.window_bytes), device=self.device)
        dist.barrier(group=group)

    def row(self, slot: int, dtype: torch.dtype) -> torch.Tensor:
        off = self.row_offset(slot)
        return self._mem[off: off + self.row_bytes].view(dtype)

    def exchange(self, step: int, lo: int, nbytes: int) -> None:
        self._check_slice(lo, nbytes)
        L = self._L
        off = self.row_offset(self.slot_of(step)) + lo
        d = L.PeerExchangeDesc()
        d.src = self.base + off
        d.bytes = nbytes
        d.world, d.rank = self.world, self.rank
        d.step = step & 0xFFFFFFFF
        d.timeout_ms = self.timeout_ms
        for p in range(self.world):
            if p != self.rank:
                d.dst[p] = self.peer_base[p] + off
                d.post[p] = self.peer_base[p] + 4 * self.rank
        d.flags = self.base
        d.status = self.base + STATUS_OFF
        L.check(self._lib.wqaa_peer_exchange(ctypes.byref(d), L.current_stream_handle(self.device)))

    def check(self) -> None:
        late = int(self._mem[STATUS_OFF: STATUS_OFF + 4].view(torch.int32).item())     # (synchronises)
        if late:
            raise PeerTimeout(f"rank {self.rank}: the post of rank {late - 1} did not arrive within {self.timeout_ms} ms")

    def close(self) -> None:
        if getattr(self, "peer_base", None):
            torch.cuda.synchronize(self.device)
            for p, ptr in enumerate(self.peer_base):
                if p != self.rank and ptr:
                    self._lib.wqaa_peer_close(ptr)
            self._lib.wqaa_peer_free(self.base)
            self.peer_base = []


def make_window(group, row_bytes: int, device, slots: int = 2, timeout_ms: int = 2000) -> PeerWindow:
    if torch.device(device).type == "cuda":
        return HipPeerWindow(group, row_bytes, device, slots, timeout_ms)
    return ShmPeerWindow(group, row_bytes, slots, timeout_ms)
