"""Host-side handle on the HIP denoiser engine (libns2vc_hip.so).

Pure ctypes + numpy: torch is optional and used only as a source of device
pointers / streams (plumbing).  Everything computes on the GPU through the C
ABI; there is no CPU path here — missing library or GPU raises ``Ns2vcError``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import Ns2vcError, PREC_BF16, PREC_F16, PREC_F32, check
from .schedule import NCOEF, SolverTable, build_table
from .spec import UNetConfig, param_spec

# operand precision of the MFMAs (include/ns2vc_hip.h): fp16 is the default 16-bit mode -- same speed and bytes as bf16,
# 7e-4 end-to-end error (inside the 1e-3 parity gate) instead of 5.7e-3
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "bf16": PREC_BF16, "fp16": PREC_F16, "f16": PREC_F16}
DEFAULT_PRECISION = "fp16"


class DevBuf:
    """A raw device allocation made through the C ABI (torch-free tests / bench)."""

    def __init__(self, nbytes: int):
        self.lib = _lib.load()
        p = C.c_void_p()
        check(self.lib.ns2vc_dev_malloc(C.byref(p), nbytes), "ns2vc_dev_malloc")
        self.ptr = p.value
        self.nbytes = nbytes

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(b.lib.ns2vc_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes), "memcpy_h2d")
        return b

    def upload(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(self.lib.ns2vc_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes), "memcpy_h2d")

    def to_numpy(self, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.lib.ns2vc_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self) -> None:
        if self.ptr:
            self.lib.ns2vc_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x) -> int:
    """Device pointer of a DevBuf / torch CUDA tensor / raw int."""
    if x is None:
        return 0
    if isinstance(x, DevBuf):
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):          # torch tensor
        if not x.is_cuda:
            raise Ns2vcError("tensor must live on the GPU: the engine has no CPU path")
        if not x.is_contiguous():
            raise Ns2vcError("tensor must be contiguous")
        return x.data_ptr()
    raise TypeError(f"cannot take a device pointer of {type(x)}")


class Stream:
    """A HIP stream: non-blocking without a mask; with ``cu_mask`` it is created by hipExtStreamCreateWithCUMask, which makes a DEFAULT
    (blocking) stream -- it synchronises implicitly with the legacy null stream, so keep other work off the null stream when overlapping.  ``cu_mask`` (an iterable of CU indices, or None): restrict the stream's kernels to those CUs
    (``hipExtStreamCreateWithCUMask``) -- how a pipeline gives the PyTorch stages a slice of the chip beside the denoiser."""

    def __init__(self, cu_mask=None):
        self.lib = _lib.load()
        p = C.c_void_p()
        if cu_mask is None:
            check(self.lib.ns2vc_stream_create(C.byref(p)), "stream_create")
        else:
            cus = sorted(set(int(c) for c in cu_mask))
            if not cus or cus[0] < 0:
                raise ValueError("cu_mask must name at least one CU")
            nw = cus[-1] // 32 + 1
            words = (C.c_uint32 * nw)()
            for c in cus:
                words[c // 32] |= 1 << (c % 32)
            check(self.lib.ns2vc_stream_create_cu_mask(C.byref(p), words, nw), "stream_create_cu_mask")
        self.ptr = p.value

    def sync(self):
        check(self.lib.ns2vc_stream_sync(self.ptr), "stream_sync")

    def __del__(self):
        try:
            if self.ptr:
                self.lib.ns2vc_stream_destroy(self.ptr)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.lib = _lib.load()
        p = C.c_void_p()
        check(self.lib.ns2vc_event_create(C.byref(p)), "event_create")
        self.ptr = p.value

    def record(self, stream) -> None:
        check(self.lib.ns2vc_event_record(self.ptr, _stream_ptr(stream)), "event_record")

    def elapsed_ms(self, end: "Event") -> float:
        ms = C.c_float()
        check(self.lib.ns2vc_event_elapsed_ms(self.ptr, end.ptr, C.byref(ms)), "event_elapsed")
        return float(ms.value)

    def __del__(self):
        try:
            if self.ptr:
                self.lib.ns2vc_event_destroy(self.ptr)
        except Exception:
            pass


def _stream_ptr(s) -> int:
    if s is None:
        return 0
    if isinstance(s, Stream):
        return s.ptr
    if isinstance(s, int):
        return s
    if hasattr(s, "cuda_stream"):       # torch.cuda.Stream
        return int(s.cuda_stream)
    raise TypeError(f"not a stream: {type(s)}")


def device_info() -> str:
    lib = _lib.load()
    buf = C.create_string_buffer(256)
    check(lib.ns2vc_device_name(buf, 256), "device_name")
    return buf.value.decode()


def set_device(index: int) -> None:
    check(_lib.load().ns2vc_set_device(int(index)), "set_device")


def xcd_round_robin() -> int:
    """1 when workgroup ids 8 apart share an XCD on the current device (probed once; what the cooperative GroupNorm prologue relies
    on, its default), 0 when not, -1 when the probe could not run."""
    n = C.c_int(0)
    check(_lib.load().ns2vc_device_xcd_round_robin(C.byref(n)), "xcd_round_robin")
    return int(n.value)


def device_count() -> int:
    lib = _lib.load()
    n = C.c_int()
    rc = lib.ns2vc_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0


class Engine:
    """One denoiser instance: weights + workspace for a (B, T, Lp) shape."""

    def __init__(self, cfg: UNetConfig = UNetConfig(), precision: str = DEFAULT_PRECISION):
        cfg.validate()
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.cfg = cfg
        self.precision = precision
        self.lib = _lib.load()
        c = _lib.UnetCfg()
        c.latent_channels = cfg.latent_channels
        c.content_channels = cfg.content_channels
        c.n_levels = len(cfg.block_out_channels)
        for i, v in enumerate(cfg.block_out_channels):
            c.block_out_channels[i] = v
        c.norm_num_groups = cfg.norm_num_groups
        c.cross_attention_dim = cfg.cross_attention_dim
        c.heads = cfg.heads
        c.layers_per_block = cfg.layers_per_block
        c.pool_heads = cfg.addition_embed_heads
        h = C.c_void_p()
        check(self.lib.ns2vc_unet_create(C.byref(c), C.byref(h)), "ns2vc_unet_create")
        self.h = h
        self.shape: Optional[Tuple[int, int, int]] = None
        self.table: Optional[SolverTable] = None
        self._weights_ready = False
        self._debug = False

    # -- weights ----------------------------------------------------------------
    def load_state_dict(self, state: Dict[str, object], strict: bool = True) -> None:
        """``state``: name -> numpy array / torch tensor (CPU or GPU), reference key names
        (optionally prefixed ``diff_model.unet.`` as in a full NS2VC checkpoint)."""
        spec = param_spec(self.cfg)
        prefix = "diff_model.unet."
        seen = set()
        for name, shape in spec.items():
            v = state.get(name, state.get(prefix + name))
            if v is None:
                if strict:
                    raise KeyError(f"missing weight {name}")
                continue
            seen.add(name)
            shp = (C.c_int64 * len(shape))(*shape)
            if hasattr(v, "data_ptr"):      # torch
                t = v.detach()
                if tuple(t.shape) != tuple(shape):
                    raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(shape)}")
                import torch  # plumbing only
                t = t.to(dtype=torch.float32).contiguous()
                check(self.lib.ns2vc_unet_load_weight(self.h, name.encode(), t.data_ptr(), shp, len(shape)), name)
            else:
                a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
                if tuple(a.shape) != tuple(shape):
                    raise ValueError(f"{name}: shape {tuple(a.shape)} != {tuple(shape)}")
                check(self.lib.ns2vc_unet_load_weight(self.h, name.encode(), a.ctypes.data, shp, len(shape)), name)
        if strict:
            extra = [k for k in state if k not in spec and not (k.startswith(prefix) and k[len(prefix):] in spec)
                     and k.startswith(("conv_", "time_embedding", "add_embedding", "down_blocks", "up_blocks", "mid_block"))]
            if extra:
                raise KeyError(f"unexpected denoiser keys: {extra[:4]}")
        check(self.lib.ns2vc_unet_finalize_weights(self.h, PRECISIONS[self.precision]), "finalize_weights")
        self._weights_ready = True
        self.shape = None

    # -- workspace ----------------------------------------------------------------
    def set_debug(self, enable: bool) -> None:
        """Keep a copy of every block output (``taps()``).  Changing it drops the plan: call ``prepare`` afterwards."""
        check(self.lib.ns2vc_unet_set_debug(self.h, int(enable)), "set_debug")
        if bool(enable) != self._debug:
            self._debug = bool(enable)
            self.shape = None

    def set_option(self, name: str, value: bool) -> None:
        """Plan options ("ln_linear", "fold_ff", "fuse_gn_gemm", "attn_optimistic", ...; include/ns2vc_hip.h).  Changing one drops
        the plan: ``prepare`` again."""
        check(self.lib.ns2vc_unet_set_option(self.h, name.encode(), int(value)), f"set_option({name})")
        self.shape = None

    def ln_ratio(self, stream=None) -> float:
        """max |mean| / std over all LayerNorm input rows since the last read-out.  The read-and-reset is enqueued on
        ``stream`` (pass the stream the evaluations ran on) and only THAT stream is synchronised.  The 16-bit modes' error
        on a LayerNorm row grows with it under the default "ln_linear" plan; see ``Denoiser`` for the automatic guard."""
        r = C.c_float()
        check(self.lib.ns2vc_unet_ln_ratio(self.h, C.byref(r), _stream_ptr(stream)), "ln_ratio")
        return float(r.value)

    def ln_ratio_post(self, stream=None) -> None:
        """enqueue the read-and-reset on ``stream`` without waiting; collect it with ``ln_ratio_poll``"""
        check(self.lib.ns2vc_unet_ln_ratio_post(self.h, _stream_ptr(stream)), "ln_ratio_post")

    def ln_ratio_poll(self) -> Optional[float]:
        """value of the last ``ln_ratio_post`` if it has completed (non-blocking), else None"""
        r, ok = C.c_float(), C.c_int()
        check(self.lib.ns2vc_unet_ln_ratio_poll(self.h, C.byref(r), C.byref(ok)), "ln_ratio_poll")
        return float(r.value) if ok.value else None

    def prepare(self, B: int, T: int, Lp: int) -> None:
        check(self.lib.ns2vc_unet_prepare(self.h, B, T, Lp), "ns2vc_unet_prepare")
        self.shape = (B, T, Lp)

    def workspace_bytes(self) -> int:
        n = C.c_size_t()
        check(self.lib.ns2vc_unet_workspace_bytes(self.h, C.byref(n)), "workspace_bytes")
        return int(n.value)

    def launches(self) -> Tuple[int, int]:
        a, b = C.c_int(), C.c_int()
        check(self.lib.ns2vc_unet_num_launches(self.h, C.byref(a), C.byref(b)), "num_launches")
        return int(a.value), int(b.value)

    # -- compute (device pointers in, device pointers out) --------------------------
    def set_condition(self, content, prompt, mask=None, stream=None) -> None:
        """content (B,256,T) fp32, prompt (B,Lp,256) fp32, mask (B,Lp) uint8/bool or None — all on the GPU."""
        check(self.lib.ns2vc_unet_set_condition(self.h, _ptr(content), _ptr(prompt), _ptr(mask), _stream_ptr(stream)), "set_condition")

    def set_content(self, content, stream=None) -> None:
        """only the content half of set_condition (the content part of conv_in)"""
        check(self.lib.ns2vc_unet_set_content(self.h, _ptr(content), _stream_ptr(stream)), "set_content")

    def set_prompt(self, prompt, mask=None, stream=None) -> None:
        """only the prompt half of set_condition (cross-attention K/V, add_embedding, mask bias)"""
        check(self.lib.ns2vc_unet_set_prompt(self.h, _ptr(prompt), _ptr(mask), _stream_ptr(stream)), "set_prompt")

    def set_mask(self, mask=None, stream=None) -> None:
        """only the keep-mask -> additive-bias conversion (None = no mask)"""
        check(self.lib.ns2vc_unet_set_mask(self.h, _ptr(mask), _stream_ptr(stream)), "set_mask")

    def forward(self, x, t, out, stream=None) -> None:
        """x (B,100,T), t (B,) fp32, out (B,100,T): one denoiser evaluation."""
        check(self.lib.ns2vc_unet_forward(self.h, _ptr(x), _ptr(t), _ptr(out), _stream_ptr(stream)), "forward")

    def load_sampler(self, solver: str, steps: int, betas: Optional[np.ndarray] = None, order: int = 2) -> SolverTable:
        table = build_table(solver, steps, betas, order)
        coef = np.ascontiguousarray(table.coef, dtype=np.float32)
        assert coef.shape == (steps, NCOEF)
        check(self.lib.ns2vc_sampler_load(self.h, steps, coef.ctypes.data_as(C.POINTER(C.c_float))), "sampler_load")
        self.table = table
        return table

    def sample(self, x_inout, use_graph: bool = True, stream=None, tail: Optional["Engine"] = None, tail_steps: int = 0) -> None:
        """x_inout (B,100,T): x_T in, sample out (in place).  NFE == steps of the loaded table.

        ``tail`` / ``tail_steps``: mixed precision -- this engine runs the first ``steps - tail_steps`` evaluations, then the
        solver state is handed to ``tail`` (normally an fp32 engine prepared for the same shape, condition and table),
        which runs the last ``tail_steps``.  The error of a sampled latent is dominated by the last evaluations
        (DPM-Solver++(2M)'s final second-order update extrapolates over a large log-SNR step, dpm_solver.py:796-831):
        two fp32 evaluations at the end take a 50-step fp16 loop from 1.7e-3 to ~2e-4 of the reference."""
        if tail is None or tail_steps <= 0:
            check(self.lib.ns2vc_sampler_run(self.h, _ptr(x_inout), int(use_graph), _stream_ptr(stream)), "sampler_run")
            return
        if self.table is None or tail.table is None or tail.table.steps != self.table.steps or \
                not np.array_equal(np.asarray(self.table.coef, dtype=np.float32), np.asarray(tail.table.coef, dtype=np.float32)):
            raise Ns2vcError("mixed-precision sampling: both engines need the SAME solver table loaded (solver, steps, order, betas)")
        n, k = self.table.steps, min(int(tail_steps), self.table.steps)
        sp = _stream_ptr(stream)
        check(self.lib.ns2vc_sampler_begin(self.h, _ptr(x_inout), sp), "sampler_begin")
        check(self.lib.ns2vc_sampler_steps(self.h, n - k, int(use_graph), sp), "sampler_steps")
        check(self.lib.ns2vc_sampler_handoff(tail.h, self.h, sp), "sampler_handoff")
        check(self.lib.ns2vc_sampler_steps(tail.h, k, int(use_graph), sp), "sampler_steps(tail)")
        check(self.lib.ns2vc_sampler_end(tail.h, _ptr(x_inout), sp), "sampler_end")

    # the loop in parts (what ``sample(tail=...)`` is made of), for callers that look at the state in mid-loop
    def sample_begin(self, x_T, stream=None) -> None:
        check(self.lib.ns2vc_sampler_begin(self.h, _ptr(x_T), _stream_ptr(stream)), "sampler_begin")

    def sample_steps(self, n: int, use_graph: bool = True, stream=None) -> None:
        check(self.lib.ns2vc_sampler_steps(self.h, int(n), int(use_graph), _stream_ptr(stream)), "sampler_steps")

    def sample_peek(self, x_out, stream=None) -> None:
        """x_e of the loop in progress (the point the NEXT evaluation is taken at) -> x_out (B,100,T); the loop goes on"""
        check(self.lib.ns2vc_sampler_peek(self.h, _ptr(x_out), _stream_ptr(stream)), "sampler_peek")

    def sample_end(self, x_out, stream=None) -> None:
        check(self.lib.ns2vc_sampler_end(self.h, _ptr(x_out), _stream_ptr(stream)), "sampler_end")

    def attn_fallbacks(self, reset: bool = True, stream=None) -> int:
        """attention workgroups whose optimistic pass (no per-tile maximum) had to be repeated by the exact pass since the last
        reset: each paid its kernel twice (scores rising > ~18 log2 units above their first 64 keys).  Waits for ``stream``."""
        n = C.c_ulonglong()
        check(self.lib.ns2vc_unet_attn_fallbacks(self.h, C.byref(n), int(reset), _stream_ptr(stream)), "attn_fallbacks")
        return int(n.value)

    def gn_coop_alone(self, reset: bool = True, stream=None) -> int:
        """workgroups of the cooperative GroupNorm prologue that waited in vain for a sibling and built every row themselves since
        the last reset (a performance counter: the values are the same either way).  Waits for ``stream``."""
        n = C.c_ulonglong()
        check(self.lib.ns2vc_unet_gn_coop_alone(self.h, C.byref(n), int(reset), _stream_ptr(stream)), "gn_coop_alone")
        return int(n.value)

    # -- profiling --------------------------------------------------------------------
    def op_info(self, which: int = 0) -> List[Tuple[str, int, float, float]]:
        """[(name, kind, algorithmic flops, algorithmic bytes)] of the per-step (0) / condition (1) plan."""
        nf, nc = self.launches()
        out = []
        name = C.create_string_buffer(256)
        kind, fl, by = C.c_int(), C.c_double(), C.c_double()
        for i in range(nc if which else nf):
            check(self.lib.ns2vc_unet_op_info(self.h, which, i, name, 256, C.byref(kind), C.byref(fl), C.byref(by)), "op_info")
            out.append((name.value.decode(), int(kind.value), float(fl.value), float(by.value)))
        return out

    def profile_forward(self, reps: int = 5, stream=None) -> np.ndarray:
        """Per-launch HIP-event timing of the per-step plan (average ms per launch over `reps` back-to-back launches)."""
        nf, _ = self.launches()
        ms = (C.c_float * nf)()
        check(self.lib.ns2vc_unet_profile_forward(self.h, ms, nf, int(reps), _stream_ptr(stream)), "profile_forward")
        return np.array(list(ms), dtype=np.float64)

    # -- debug taps -----------------------------------------------------------------
    def taps(self) -> Dict[str, np.ndarray]:
        out: Dict[str, np.ndarray] = {}
        n = self.lib.ns2vc_unet_num_taps(self.h)
        name = C.create_string_buffer(256)
        rows, cols = C.c_int(), C.c_int()
        for i in range(n):
            check(self.lib.ns2vc_unet_tap_info(self.h, i, name, 256, C.byref(rows), C.byref(cols)), "tap_info")
            a = np.empty((rows.value, cols.value), dtype=np.float32)
            check(self.lib.ns2vc_unet_tap_read(self.h, i, a.ctypes.data), "tap_read")
            out[name.value.decode()] = a
        return out

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.ns2vc_unet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sync() -> None:
    check(_lib.load().ns2vc_dev_sync(), "dev_sync")
