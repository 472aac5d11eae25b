"""Target detection and the gfx950 device descriptor.

Reference counterparts: `bitblas/utils/target_detector.py:82-105` (`auto_detect_nvidia_target`
shells out to nvidia-smi and fuzzy-matches a TVM target tag) and `bitblas/base/arch/cdna.py:14-35`
(the `CDNA` TileDevice, which hard-codes `bandwidth=[1300, 14000]`, `reg_cap=32768`).  The public
name `auto_detect_nvidia_target` is kept because `Linear` and user code call it; on this backend it
reports the HIP device.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List

DEFAULT_TARGET = "hip -mcpu=gfx950"


@dataclass(frozen=True)
class CDNA4:
    """What the tile selector needs to know about MI355X (numbers: MI355X_MICROARCH notes)."""
    name: str = "gfx950"
    platform: str = "CDNA"
    compute_units: int = 256
    xcds: int = 8
    simds_per_cu: int = 4
    warp_size: int = 64
    max_waves_per_cu: int = 32
    lds_bytes_per_cu: int = 160 * 1024
    vgprs_per_simd_lane: int = 512
    l2_bytes_per_xcd: int = 4 << 20
    infinity_cache_bytes: int = 256 << 20
    hbm_bytes: int = 288 << 30
    hbm_peak_gbs: float = 8000.0           # spec; ~6300 achievable on a streaming copy
    mfma_f16_dense_tflops: float = 2500.0
    mfma_i8_dense_tops: float = 5000.0
    mfma_fp8_dense_tflops: float = 5000.0
    max_clock_mhz: int = 2400
    # fields the reference's TileDevice exposes and callers occasionally read
    smem_cap: int = 160 * 1024
    reg_cap: int = 512 * 64 * 4
    sm_partition: int = 4
    bandwidth: List[int] = field(default_factory=lambda: [8000, 34500])

    def get_avaliable_tensorintrin_shapes(self):  # sic - reference spelling
        return [[16, 16, 32], [32, 32, 16]]


def auto_detect_nvidia_target(gpu_id: int = 0) -> str:
    """Return the target tag of the visible accelerator.

    Honours `BITBLAS_TARGET` / `TVM_TARGET` like the reference's auto-detect; otherwise asks the
    HIP runtime (through torch) for the gcn arch and falls back to gfx950 when no device is
    visible (e.g. the CPU-only build container)."""
    for var in ("BITBLAS_TARGET", "TVM_TARGET"):
        if os.environ.get(var):
            return os.environ[var]
    try:
        import torch
        if torch.cuda.is_available():
            arch = torch.cuda.get_device_properties(gpu_id).gcnArchName.split(":")[0]
            return f"hip -mcpu={arch}"
    except Exception:  # pragma: no cover - detection must never break import
        pass
    return DEFAULT_TARGET


auto_detect_target = auto_detect_nvidia_target


def get_arch(target=None):
    """Target tag -> device descriptor (reference: bitblas/base/arch/__init__.py:11-23)."""
    t = str(target or DEFAULT_TARGET)
    if t.startswith("hip") or "gfx" in t:
        return CDNA4(name=t.split("=")[-1] if "=" in t else "gfx950")
    if t.startswith("cuda") or t.startswith("nvidia"):
        raise ValueError("bitblas_amd only ships gfx950 kernels; CUDA targets are not supported")
    raise ValueError(f"Unsupported target {t!r}: only hip (gfx950) is supported")
