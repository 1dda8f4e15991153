"""Batched voice-conversion service on top of the denoiser engine -- SURVEY 8(f) rank 3.

The reference's ``Svc.infer`` (``inference/infer_tool.py:189-206``) converts ONE audio segment per call (batch 1,
``infer.py:99-140`` loops over the slicer's segments).  Here segments are converted in batches:

* segments are grouped by (latent length T, prompt length Lp) and never padded.  The reference applies no masking to
  padded LATENT frames (no self-attention / GroupNorm mask, ``model.py:411`` is commented out), and although padded
  PROMPT frames are masked out of cross-attention (``encoder_attention_mask``) they still enter both attention-pooled
  embeddings (``ref_enc`` ``model.py:362`` and the UNet's ``add_embedding``, which take no mask) -- so padding either
  side changes a segment's result (measured: 0.2-0.5 relative on the latent for a prompt padded 40 -> 64).  Grouping
  EQUAL shapes keeps every segment's result what the reference's batch-1 call gives, to within the precision's rounding
  noise.  A conversion job normally shares ONE reference clip over all its segments (``infer.py:92-122``: the segment loop sits inside the loop over reference clips), so Lp rarely
  splits a group.
* each group runs ``PreModel.infer`` -> ``Denoiser.sample`` -> ``decode_fn`` through ``OverlappedPipeline``: the
  PyTorch-ROCm front / back end of group k+1 / k-1 overlaps the HIP denoiser of group k on their own streams.

``decode_fn(latent (B, 100, T)) -> audio (B, samples)`` is the vocoder (Vocos in the reference, ``model.py:689-691``;
not a dependency of this repository: pass ``vocos.decode``); with ``decode_fn=None`` the latents are returned.
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch

from .frontend import PreModel
from .pipeline import Denoiser, OverlappedPipeline


@dataclass
class Segment:
    """one unit of work: ContentVec features (256, T) already repeat-expanded to the mel frame rate
    (``utils.repeat_expand_2d``, ``infer_tool.py:158-168``) and the reference mel (100, Lp) (``infer_tool.py:170-182``)"""
    content: torch.Tensor
    refer: torch.Tensor
    tag: object = None


def segment_from_audio(content_encoder, wav16k: torch.Tensor, samples_at_target_rate: int, refer_mel: torch.Tensor, hop: int = 256,
                       tag: object = None, autocast=None) -> Segment:
    """``Svc.get_unit_f0_code`` minus the file I/O (``infer_tool.py:141-182``): the 16 kHz copy of ONE source segment -> ContentVec
    features (``ns2vc_amd.contentvec.ContentVec`` or anything with its ``content(wav16k, frames)``) stretched to the segment's latent
    frame count ``len(wav at 24 kHz) // hop`` (what the reference takes from its f0 track, ``utils.py:160``), plus the prompt mel."""
    frames = int(samples_at_target_rate) // hop
    return Segment(content=content_encoder.content(wav16k, frames, autocast=autocast)[0], refer=refer_mel, tag=tag)


class GroupedConverter:
    def __init__(self, pre_model: PreModel, denoiser: Denoiser, decode_fn: Optional[Callable] = None, max_batch: int = 32,
                 solver: str = "unipc", steps: int = 30, order: int = 2, seed: int = 1234):
        self.pre, self.den, self.decode = pre_model, denoiser, decode_fn
        self.max_batch, self.seed = max_batch, seed
        self.kw = dict(solver=solver, steps=steps, order=order)

    def plan(self, segments: Sequence[Segment]) -> List[List[int]]:
        """indices of `segments` grouped by (latent length, prompt length), groups of at most ``max_batch``, longest first"""
        by_len: Dict[tuple, List[int]] = defaultdict(list)
        for i, s in enumerate(segments):
            by_len[(int(s.content.shape[-1]), int(s.refer.shape[-1]))].append(i)
        groups = []
        for key in sorted(by_len, reverse=True):
            idx = by_len[key]
            groups += [idx[k:k + self.max_batch] for k in range(0, len(idx), self.max_batch)]
        return groups

    def convert(self, segments: Sequence[Segment]) -> List[torch.Tensor]:
        """returns one tensor per segment, in input order: audio (samples,) with a ``decode_fn``, else the latent (100, T)"""
        dev = next(self.pre.parameters()).device
        groups = self.plan(segments)

        def pre_fn(idx):
            T, Lp = int(segments[idx[0]].content.shape[-1]), int(segments[idx[0]].refer.shape[-1])
            c = torch.stack([segments[i].content.to(dev, torch.float32) for i in idx])
            refer = torch.stack([segments[i].refer.to(dev, torch.float32) for i in idx])
            content, prompt, mask = self.pre.infer(c, refer, torch.full((len(idx),), T, device=dev), torch.full((len(idx),), Lp, device=dev))
            # x_T per SEGMENT (seeded by its position in the input), so a segment's result does not depend on its group
            noise = torch.stack([torch.randn((self.den.cfg.latent_channels, T), generator=torch.Generator().manual_seed(self.seed + i)) for i in idx]).to(dev)
            return {"content": content, "prompt": prompt, "prompt_mask": mask, "noise": noise}

        def post_fn(latent, idx):
            return latent if self.decode is None else self.decode(latent)

        outs = OverlappedPipeline(self.den, pre_fn, post_fn, **self.kw).run(groups)
        result: List[Optional[torch.Tensor]] = [None] * len(segments)
        for idx, o in zip(groups, outs):
            for b, i in enumerate(idx):
                result[i] = o[b]
        return result
