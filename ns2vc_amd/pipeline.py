"""High-level denoiser sampling API on torch CUDA tensors (torch = memory/stream plumbing).

This is what a maintainer calls from ``NaturalSpeech2.sample`` instead of the
reference's per-step Python loop (``model.py:620-687``): the step-invariant
condition work is hoisted once, the N-step DPM-Solver++ / UniPC loop replays one
captured hipGraph per step, and nothing synchronises with the host until the end.
"""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import numpy as np

from . import dist as _dist
from .engine import DEFAULT_PRECISION, Engine
from .schedule import linear_betas
from .spec import UNetConfig


class Denoiser:
    """``precision``: "fp16" (default: 16-bit MFMA operands, 8e-4 vs the reference fp32 path -- inside the 1e-3 parity
    bar), "fp32" (exact-fp32 MFMA, 1e-6) or "bf16" (same speed as fp16, 6.5e-3; kept for range-critical checkpoints).

    ``ln_guard``: LayerNorm by linearity (the default plan) lets the 16-bit modes round a LayerNorm's RAW input before
    centring, so their error on a row grows with |mean| / std of that row.  After the FIRST evaluation with a new set
    of weights the engine's measured maximum of that ratio is read once (one host sync); above ``ln_guard`` the plan is
    switched to explicit LayerNorm passes (``ln_linear`` 0: ~4 % slower, immune) and the evaluation is repeated.
    ``None`` disables the check."""

    def __init__(self, state: Dict[str, object], cfg: UNetConfig = UNetConfig(), precision: str = DEFAULT_PRECISION,
                 betas: Optional[np.ndarray] = None, ln_guard: Optional[float] = 8.0):
        self.cfg = cfg
        self.engine = Engine(cfg, precision=precision)
        self.engine.load_state_dict(state)
        self.betas = linear_betas() if betas is None else np.asarray(betas, dtype=np.float32)
        self._shape = None
        self._table_key = None
        self.ln_guard = ln_guard if precision not in ("fp32", "f32") else None
        self.ln_ratio_seen: Optional[float] = None
        self._ln_checked = False

    def _guard(self, redo):
        """first-call health check of the LayerNorm-by-linearity plan (see the class docstring)"""
        if self._ln_checked or self.ln_guard is None:
            return None
        self._ln_checked = True
        self.ln_ratio_seen = self.engine.ln_ratio()
        if self.ln_ratio_seen <= self.ln_guard:
            return None
        warnings.warn(f"LayerNorm inputs with |mean|/std up to {self.ln_ratio_seen:.1f} (> {self.ln_guard}): switching the "
                      f"{self.engine.precision} engine to explicit LayerNorm passes (ln_linear=0)")
        self.engine.set_option("ln_linear", False)
        self._shape = None
        return redo()

    def _prepare(self, B: int, T: int, Lp: int) -> None:
        if self._shape != (B, T, Lp):
            import torch
            torch.cuda.synchronize()
            self.engine.prepare(B, T, Lp)
            self._shape = (B, T, Lp)

    def _table(self, solver: str, steps: int, order: int) -> None:
        key = (solver, steps, order)
        if self._table_key != key:
            self.engine.load_sampler(solver, steps, self.betas, order)
            self._table_key = key

    def denoise(self, x, t, content, prompt, prompt_mask=None):
        """One evaluation: x (B,100,T), t (B,), content (B,256,T), prompt (B,Lp,256), mask (B,Lp) bool -> x0_pred."""
        import torch
        B, _, T = x.shape
        self._prepare(B, T, prompt.shape[1])
        s = torch.cuda.current_stream(x.device)
        mask = None if prompt_mask is None else prompt_mask.to(torch.uint8).contiguous()
        self.engine.set_condition(content.float().contiguous(), prompt.float().contiguous(), mask, stream=s)
        out = torch.empty_like(x, dtype=torch.float32)
        self.engine.forward(x.float().contiguous(), t.float().contiguous(), out, stream=s)
        redone = self._guard(lambda: self.denoise(x, t, content, prompt, prompt_mask))
        return out if redone is None else redone

    def sample(self, content, prompt, prompt_mask=None, noise=None, solver: str = "unipc", steps: int = 20, order: int = 2,
               use_graph: bool = True, generator=None):
        """content (B,256,T), prompt (B,Lp,256), mask (B,Lp) bool; ``noise`` (B,100,T) = x_T (drawn with
        torch.randn like model.py:635 if None).  Returns the sampled latent (B,100,T) fp32."""
        import torch
        B, _, T = content.shape
        dev = content.device
        self._prepare(B, T, prompt.shape[1])
        self._table(solver, steps, order)
        if noise is None:
            noise = torch.randn((B, self.cfg.latent_channels, T), device=dev, generator=generator)
        x = noise.to(device=dev, dtype=torch.float32).contiguous().clone()
        s = torch.cuda.current_stream(dev)
        mask = None if prompt_mask is None else prompt_mask.to(device=dev, dtype=torch.uint8).contiguous()
        self.engine.set_condition(content.float().contiguous(), prompt.float().contiguous(), mask, stream=s)
        self.engine.sample(x, use_graph=use_graph, stream=s)
        redone = self._guard(lambda: self.sample(content, prompt, prompt_mask, noise, solver, steps, order, use_graph))
        return x if redone is None else redone

    def sample_sharded(self, content, prompt, prompt_mask, noise, **kw):
        """Data-parallel: every rank receives the GLOBAL batch description, runs its contiguous slice and the
        finished latents are all-gathered (RCCL).  The noise is drawn for the global batch and sliced, so an utterance's
        result does not depend on the world size beyond the precision's rounding noise (fp32: ~1e-6; 16-bit: a shard of
        3 and a shard of 2 round differently, ~7e-4 -- tests/test_dropin_gpu.py::test_sample_sharded_rccl_*)."""
        import torch
        import torch.distributed as td
        rank = td.get_rank() if td.is_initialized() else 0
        world = td.get_world_size() if td.is_initialized() else 1
        n = content.shape[0]
        lo, hi = _dist.shard_range(n, rank, world)
        if hi == lo:      # fewer utterances than ranks: this rank has nothing to denoise but must still join the collective
            local = torch.zeros((0, self.cfg.latent_channels, content.shape[2]), dtype=torch.float32, device=content.device)
        else:
            pm = None if prompt_mask is None else prompt_mask[lo:hi]
            local = self.sample(content[lo:hi], prompt[lo:hi], pm, noise[lo:hi], **kw)
        return _dist.gather_latents(local, n)


class OverlappedPipeline:
    """Three-stage utterance-batch pipeline on three HIP streams: the PyTorch-ROCm front end (ContentVec / ``Pre_model.infer``,
    reference ``model.py:359-376``) and back end (``vocos.decode``, ``model.py:689-696``) overlap the denoiser instead of
    running before / after it (BASELINE.json north_star):

        pre(k+1)   |   denoise(k)   |   post(k-1)

    ``pre_fn(item)`` runs on the front-end stream and returns a dict with torch CUDA tensors ``content`` (B,256,T),
    ``prompt`` (B,Lp,256) and optionally ``prompt_mask`` (B,Lp) bool and ``noise`` (B,100,T); ``post_fn(latent, item)``
    runs on the back-end stream with the sampled latent (B,100,T) fp32.  Ordering is by events only -- the host never
    blocks until the end of ``run`` -- and the denoiser still replays one captured hipGraph per step on its own stream.
    """

    def __init__(self, denoiser: Denoiser, pre_fn, post_fn, solver: str = "unipc", steps: int = 20, order: int = 2,
                 use_graph: bool = True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("OverlappedPipeline needs a ROCm device (there is no CPU path)")
        self.denoiser, self.pre_fn, self.post_fn = denoiser, pre_fn, post_fn
        self.kw = dict(solver=solver, steps=steps, order=order, use_graph=use_graph)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.s_pre, self.s_den, self.s_post = (torch.cuda.Stream(dev) for _ in range(3))

    def _launch_pre(self, item):
        import torch
        with torch.cuda.stream(self.s_pre):
            cond = self.pre_fn(item)
            ev = torch.cuda.Event()
            ev.record(self.s_pre)
        return item, cond, ev

    def run(self, items):
        """Process an iterable of work items; returns ``[post_fn(latent_k, item_k) for k]`` (all streams drained)."""
        import torch
        it = iter(items)
        first = next(it, None)
        cur = self._launch_pre(first) if first is not None else None
        results = []
        while cur is not None:
            item, cond, ev_pre = cur
            nxt = next(it, None)
            nxt_pre = self._launch_pre(nxt) if nxt is not None else None     # front end of batch k+1 under denoise(k)
            with torch.cuda.stream(self.s_den):
                self.s_den.wait_event(ev_pre)
                for v in cond.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(self.s_den)
                latent = self.denoiser.sample(cond["content"], cond["prompt"], cond.get("prompt_mask"), cond.get("noise"), **self.kw)
                ev_den = torch.cuda.Event()
                ev_den.record(self.s_den)
            with torch.cuda.stream(self.s_post):
                self.s_post.wait_event(ev_den)
                latent.record_stream(self.s_post)
                results.append(self.post_fn(latent, item))                   # back end of batch k under denoise(k+1)
            cur = nxt_pre
        self.s_post.synchronize()
        self.s_den.synchronize()
        return results
