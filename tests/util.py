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


def f16_round(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even to IEEE binary16 (overflow -> inf), returned as float32"""
    with np.errstate(over="ignore"):
        return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


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


# ---- shared with tests/golden/make_golden_v2.py (the generator and the tests must build identical data) ----
def g7_summary(y: np.ndarray) -> dict:
    """what is stored of a (1, 100, 2813) output instead of its 1.1 MB"""
    T = y.shape[-1]
    mid = (T // 2) // 64 * 64
    return {"head": y[:, :, :256].copy(), "mid": y[:, :, mid:mid + 256].copy(), "tail": y[:, :, T - 256:].copy(), "mid_start": np.array([mid]),
            "chan_sum": y.astype(np.float64).sum(-1), "chan_sq": (y.astype(np.float64) ** 2).sum(-1),
            "frame_sum": y.astype(np.float64).sum(1), "frame_sq": (y.astype(np.float64) ** 2).sum(1)}


def tte_state(prefix: str, dim: int, out_dim: int) -> dict:
    """procedural parameters of a TextTimeEmbedding(dim, out_dim, heads), reference parameter names (torch tensors)"""
    import torch
    from ns2vc_amd.weights import hash_normal
    s = dim ** -0.5
    sd = {"norm1.weight": 1.0 + 0.1 * hash_normal(prefix + "n1w", (dim,)), "norm1.bias": 0.1 * hash_normal(prefix + "n1b", (dim,)),
          "pool.positional_embedding": s * hash_normal(prefix + "pos", (1, dim)),
          "proj.weight": s * hash_normal(prefix + "pw", (out_dim, dim)), "proj.bias": 0.1 * hash_normal(prefix + "pb", (out_dim,)),
          "norm2.weight": 1.0 + 0.1 * hash_normal(prefix + "n2w", (out_dim,)), "norm2.bias": 0.1 * hash_normal(prefix + "n2b", (out_dim,))}
    for p in ("k_proj", "q_proj", "v_proj"):
        sd[f"pool.{p}.weight"] = s * hash_normal(prefix + p + "w", (dim, dim))
        sd[f"pool.{p}.bias"] = 0.1 * hash_normal(prefix + p + "b", (dim,))
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()}


def procedural_params(shapes, tag: str) -> dict:
    """deterministic parameters for a {name: shape} list (torch tensors): matrices ~ N(0, 1/fan_in), norm weights ~ 1,
    biases / vectors ~ 0.1 N(0, 1); the same integer-hash generator as the denoiser's procedural weights"""
    import torch
    from ns2vc_amd.weights import hash_normal
    out = {}
    for name, shape in shapes:
        shape = tuple(shape)
        v = hash_normal(f"{tag}.{name}", shape)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:])) if "conv.weight" not in name else shape[0] * shape[1]
            v = v / np.sqrt(max(fan_in, 1))
        elif "norm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * v
        else:
            v = 0.1 * v
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out
