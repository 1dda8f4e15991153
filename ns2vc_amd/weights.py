"""Procedural (hash-based) weights and checkpoint I/O for the denoiser.

There is no trained NS2VC checkpoint offline and a random-init state dict is
252 MB, so golden vectors are pinned with a *procedural* state dict: every
value is a pure function of (seed, parameter name, flat index), computed with
integer arithmetic in numpy only.  The same function is evaluated in the
container that imports the reference (to generate ``tests/golden``) and on the
GPU box (to feed the HIP engine), so nothing large has to be committed.

Checkpoint layout kept compatible with the reference (``model.py:808-829``,
``inference/infer_tool.py:24-29``): ``{'step': int, 'model': state_dict}`` with
the UNet under the ``diff_model.unet.`` prefix.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import numpy as np

from .spec import UNetConfig, param_spec

_M1 = np.uint64(0xFF51AFD7ED558CCD)
_M2 = np.uint64(0xC4CEB9FE1A85EC53)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x: np.ndarray) -> np.ndarray:
    """murmur3 fmix64 on a uint64 array (wrapping arithmetic)."""
    x = x.copy()
    x ^= x >> np.uint64(33)
    x *= _M1
    x ^= x >> np.uint64(33)
    x *= _M2
    x ^= x >> np.uint64(33)
    return x


def hash_uniform(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float32 values in [-1, 1), a pure function of (seed, name, index)."""
    key = np.uint64((zlib.crc32(name.encode()) & 0xFFFFFFFF) | ((seed & 0xFFFFFFFF) << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * _GOLD + _mix64(np.array([key], dtype=np.uint64))[0]
        h = _mix64(idx)
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))   # 24-bit mantissa, exact in f32
    return (2.0 * u - 1.0).astype(np.float32)


def _fan_in(shape: Tuple[int, ...]) -> int:
    n = 1
    for s in shape[1:]:
        n *= s
    return n


def procedural_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, fan_in_hint: int | None = None) -> np.ndarray:
    n = int(np.prod(shape))
    u = hash_uniform(name, n, seed)
    leaf = name.rsplit(".", 2)
    is_norm = any(tok in name for tok in (".norm", "conv_norm_out", "norm1.", "norm2.", "norm3."))
    if name.endswith("positional_embedding"):
        v = u * np.float32(1.0 / np.sqrt(shape[-1]))
    elif is_norm and name.endswith(".weight"):
        v = np.float32(1.0) + np.float32(0.1) * u
    elif is_norm and name.endswith(".bias"):
        v = np.float32(0.1) * u
    elif len(shape) >= 2:
        v = u * np.float32(1.0 / np.sqrt(_fan_in(shape)))
    else:  # bias of a conv/linear: torch-default-like bound 1/sqrt(fan_in of its weight)
        fi = fan_in_hint if fan_in_hint else max(shape[0], 1)
        v = u * np.float32(1.0 / np.sqrt(fi))
    del leaf
    return v.reshape(shape)


def procedural_state_dict(cfg: UNetConfig = UNetConfig(), seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Full UNet state dict (numpy float32) for ``cfg``; deterministic everywhere."""
    spec = param_spec(cfg)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in spec.items():
        hint = None
        if name.endswith(".bias"):
            w = spec.get(name[:-5] + ".weight")
            if w is not None and len(w) >= 2:
                hint = _fan_in(w)
        out[name] = procedural_tensor(name, shape, seed, hint)
    return out


def hash_normal(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """Deterministic N(0,1)-like float32 tensor (sum of 4 uniforms, variance-matched);
    used for synthetic bench inputs where bit-identical data on every rank matters
    more than exact Gaussianity."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float32)
    for k in range(4):
        acc += hash_uniform(f"{name}#{k}", n, seed)
    return (acc * np.float32(np.sqrt(3.0 / 4.0))).reshape(shape)


# ----------------------------------------------------------------------------
# checkpoint helpers (torch imported lazily: plumbing only)
# ----------------------------------------------------------------------------
UNET_PREFIX = "diff_model.unet."


def unet_state_from_checkpoint(ckpt: Dict, cfg: UNetConfig = UNetConfig()) -> "OrderedDict[str, object]":
    """Extract the denoiser's tensors from a reference checkpoint dict
    (``{'step','model'}``, keys prefixed ``diff_model.unet.``) or from a bare
    UNet state dict.  Raises KeyError listing what is missing."""
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict) else ckpt
    spec = param_spec(cfg)
    out = OrderedDict()
    missing = []
    for name in spec:
        if UNET_PREFIX + name in sd:
            out[name] = sd[UNET_PREFIX + name]
        elif name in sd:
            out[name] = sd[name]
        else:
            missing.append(name)
    if missing:
        raise KeyError(f"{len(missing)} denoiser tensors missing from checkpoint, e.g. {missing[:4]}")
    for name, shape in spec.items():
        if tuple(out[name].shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {tuple(out[name].shape)} != expected {tuple(shape)}")
    return out


def load_checkpoint(path: str, cfg: UNetConfig = UNetConfig()):
    import torch

    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return unet_state_from_checkpoint(ckpt, cfg)


def save_checkpoint(path: str, unet_state: Dict[str, object], step: int = 0, extra: Dict[str, object] | None = None) -> None:
    """Write a reference-layout checkpoint holding (at least) the denoiser."""
    import torch

    model = OrderedDict()
    for k, v in unet_state.items():
        model[UNET_PREFIX + k] = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
    if extra:
        model.update(extra)
    torch.save({"step": int(step), "model": model}, path)
