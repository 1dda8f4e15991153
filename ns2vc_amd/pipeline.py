"""High-level denoiser sampling API on torch CUDA tensors (torch = memory/stream plumbing).

This is what a maintainer calls from ``NaturalSpeech2.sample`` instead of the
reference's per-step Python loop (``model.py:620-687``): the step-invariant
condition work is hoisted once, the N-step DPM-Solver++ / UniPC loop replays one
captured hipGraph per step, and nothing synchronises with the host until the end.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import dist as _dist
from .engine import Engine
from .schedule import linear_betas
from .spec import UNetConfig


class Denoiser:
    def __init__(self, state: Dict[str, object], cfg: UNetConfig = UNetConfig(), precision: str = "bf16",
                 betas: Optional[np.ndarray] = None):
        self.cfg = cfg
        self.engine = Engine(cfg, precision=precision)
        self.engine.load_state_dict(state)
        self.betas = linear_betas() if betas is None else np.asarray(betas, dtype=np.float32)
        self._shape = None
        self._table_key = None

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
        return out

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
        return x

    def sample_sharded(self, content, prompt, prompt_mask, noise, **kw):
        """Data-parallel: every rank receives the GLOBAL batch description, runs its contiguous slice and the
        finished latents are all-gathered (RCCL).  Results are identical for any world size because the noise
        is drawn for the global batch and sliced."""
        import torch.distributed as td
        rank = td.get_rank() if td.is_initialized() else 0
        world = td.get_world_size() if td.is_initialized() else 1
        n = content.shape[0]
        lo, hi = _dist.shard_range(n, rank, world)
        pm = None if prompt_mask is None else prompt_mask[lo:hi]
        local = self.sample(content[lo:hi], prompt[lo:hi], pm, noise[lo:hi], **kw)
        return _dist.gather_latents(local, n)
