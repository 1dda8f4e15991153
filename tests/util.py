"""numpy references and helpers shared by the tests (test-side code, not product)."""
from __future__ import annotations

import numpy as np


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def bf16_round(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def silu(x):
    return x / (1.0 + np.exp(-x))


def gelu_erf(x):
    from math import erf
    return 0.5 * x * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))


def gather_rows(a: np.ndarray, B: int, Tin: int, Tout: int, taps: int, tmode: int) -> np.ndarray:
    """a: [B, Tin, C] -> [B, Tout, taps, C] with the conv's zero padding / stride / nearest-upsample indexing."""
    C = a.shape[-1]
    out = np.zeros((B, Tout, taps, C), dtype=a.dtype)
    for t in range(Tout):
        for tap in range(taps):
            if tmode == 0:
                tt = t + tap - taps // 2
                ok = 0 <= tt < Tin
            elif tmode == 1:
                tt = 2 * t + tap - 1
                ok = 0 <= tt < Tin
            else:
                u = t + tap - 1
                ok = 0 <= u < Tout
                tt = min(u >> 1, Tin - 1)
            if ok:
                out[:, t, tap, :] = a[:, tt, :]
    return out
