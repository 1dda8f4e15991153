"""Conditioning front end of NS2VC (``Pre_model``) on PyTorch-ROCm -- SURVEY 8(f) rank 1.

The north star keeps the once-per-utterance conditioning encoders on PyTorch-ROCm and stream-overlaps them with the
HIP denoiser (``ns2vc_amd.pipeline.OverlappedPipeline``).  This module restates the reference's
``Pre_model.infer`` (``model.py:328-376``: ``ref_enc`` = ``TextTimeEmbedding(100, 100, 1)``, ``PromptEncoder``
``model.py:146-190``, ``PhoneEncoder`` ``model.py:98-144``; layers = ``EncSALayer`` ``operations.py:784-822`` with
fairseq-style ``MultiheadAttention`` ``:304-414`` and the k=9 conv feed-forward ``TransformerFFNLayer`` ``:644-692``)
with the reference's parameter names, so the ``pre_model.*`` part of a reference checkpoint loads strict, but in the
shape the GPU wants:

* batch-first tensors end to end (the reference runs (T, B, C) and permutes around every op);
* the nine shifted ``Linear`` layers of the conv feed-forward as ONE ``F.conv1d`` (k = 9) -- the weights are stacked on
  the fly from the reference's ``ffn_1.{0..8}`` parameters, the ``9 ** -0.5`` scale folded in (and the reference's quirk
  kept: its tap 0 reads the unshifted input, ``operations.py:676``);
* attention through ``F.scaled_dot_product_attention`` with the key-padding mask as a boolean mask (the reference asks
  ``F.multi_head_attention_forward`` for averaged attention weights it never uses);
* outputs already in the denoiser's layouts: content (B, 256, T), prompt (B, Lp, 256), prompt_mask (B, Lp).

Inference only (dropout is the identity); verified against outputs of the reference's own ``Pre_model`` on
procedural parameters (tests/golden/make_golden_v2.py -> golden_v2.npz ``g10.*``; tests/test_cpu.py).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from unet1d.embeddings import TextTimeEmbedding


class _ConvTBC(nn.Module):
    """parameter container with the reference's ConvTBC layout: weight (k, c_in, c_out), bias (c_out)"""

    def __init__(self, c_in: int, c_out: int, k: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(k, c_in, c_out).normal_(0, (4.0 / (k * c_in)) ** 0.5))
        self.bias = nn.Parameter(torch.zeros(c_out))


class ConvLayer(nn.Module):
    """LayerNorm -> k=1 ConvTBC (= Linear); padded frames are zeroed before the norm (model.py:78-97)"""

    def __init__(self, c_in: int, c_out: int):
        super().__init__()
        self.layer_norm = nn.LayerNorm(c_in)
        self.conv = _ConvTBC(c_in, c_out, 1)

    def forward(self, x: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:          # x (B, T, C), keep (B, T, 1) float
        return F.linear(self.layer_norm(x * keep), self.conv.weight[0].t(), self.conv.bias)


class _SelfAttention(nn.Module):
    def __init__(self, c: int, heads: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * c, c))
        nn.init.xavier_uniform_(self.in_proj_weight)
        self.out_proj = nn.Linear(c, c, bias=False)
        self.heads = heads

    def forward(self, x: torch.Tensor, key_keep: torch.Tensor) -> torch.Tensor:       # x (B, T, C), key_keep (B, T) bool
        B, T, C = x.shape
        H = self.heads
        q, k, v = F.linear(x, self.in_proj_weight).view(B, T, 3, H, C // H).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=key_keep[:, None, None, :])
        return self.out_proj(o.transpose(1, 2).reshape(B, T, C))


class _ConvFFN(nn.Module):
    """TransformerFFNLayer with kernel_size 9, 'SAME' padding: sum_i Linear_i(x[t + i - 4]) * 9^-0.5 -> ReLU -> Linear"""

    def __init__(self, c: int, k: int = 9):
        super().__init__()
        self.ffn_1 = nn.ModuleList([nn.Linear(c, 4 * c, bias=(i == 0)) for i in range(k)])
        self.ffn_2 = nn.Linear(4 * c, c)
        self.k = k

    def forward(self, x: torch.Tensor) -> torch.Tensor:                                # (B, T, C)
        # tap i reads x[t + i - k//2] -- except tap 0, which the reference feeds the UNSHIFTED x
        # (`shifted = padded[i:T+i] if i else x`, operations.py:676): its weight joins the centre tap, offset -k//2 is empty
        taps = [torch.zeros_like(self.ffn_1[0].weight)] + [l.weight for l in self.ffn_1[1:]]
        taps[self.k // 2] = taps[self.k // 2] + self.ffn_1[0].weight
        w = torch.stack(taps, dim=-1) * self.k ** -0.5                                  # (4C, C, k)
        h = F.conv1d(x.transpose(1, 2), w, self.ffn_1[0].bias * self.k ** -0.5, padding=self.k // 2)
        return self.ffn_2(F.relu(h).transpose(1, 2))


class _EncSALayer(nn.Module):
    def __init__(self, c: int, heads: int = 8):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(c)
        self.self_attn = _SelfAttention(c, heads)
        self.layer_norm2 = nn.LayerNorm(c)
        self.ffn = _ConvFFN(c)

    def forward(self, x, keep, key_keep):
        x = (x + self.self_attn(self.layer_norm1(x), key_keep)) * keep
        return (x + self.ffn(self.layer_norm2(x))) * keep


class _Layer(nn.Module):                      # the reference wraps every layer as TransformerEncoderLayer(...).op
    def __init__(self, c: int):
        super().__init__()
        self.op = _EncSALayer(c)

    def forward(self, x, keep, key_keep):
        return self.op(x, keep, key_keep)


class _Encoder(nn.Module):
    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, n_layers: int, speaker: bool):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(hidden_channels) for _ in range(n_layers)])
        self.pre = ConvLayer(in_channels, hidden_channels)
        self.out_proj = ConvLayer(hidden_channels, out_channels)
        self.layer_norm = nn.LayerNorm(out_channels)
        if speaker:
            self.spk_proj = nn.Conv1d(100, hidden_channels, 1)

    def forward(self, x_bct: torch.Tensor, lengths: torch.Tensor, g: torch.Tensor = None) -> torch.Tensor:
        """x (B, C, T) -> (B, T, out_channels); frames at or beyond `lengths` come out as zeros"""
        if g is not None:
            x_bct = x_bct + self.spk_proj(g)
        x = x_bct.transpose(1, 2)
        key_keep = torch.arange(x.shape[1], device=x.device)[None, :] < lengths[:, None]
        keep = key_keep[:, :, None].to(x.dtype)
        x = self.pre(x, keep) * keep
        for layer in self.layers:
            x = layer(x, keep, key_keep)
        return self.layer_norm(self.out_proj(x, keep)) * keep


class PreModel(nn.Module):
    """``Pre_model`` of the reference (model.py:328-376); ``cfg`` = the reference's config.json dict."""

    def __init__(self, cfg: Dict):
        super().__init__()
        pe, pr = cfg["phoneme_encoder"], cfg["prompt_encoder"]
        self.phoneme_encoder = _Encoder(pe["in_channels"], pe["hidden_channels"], pe["out_channels"], pe["n_layers"], speaker=True)
        self.prompt_encoder = _Encoder(pr["in_channels"], pr["hidden_channels"], pr["out_channels"], pr["n_layers"], speaker=False)
        self.ref_enc = TextTimeEmbedding(100, 100, 1)

    @torch.no_grad()
    def infer(self, c_padded: torch.Tensor, refer_padded: torch.Tensor, lengths: torch.Tensor, refer_lengths: torch.Tensor,
              autocast: Optional[torch.dtype] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """c_padded (B, 256, T) ContentVec features, refer_padded (B, 100, Lp) reference mel, lengths / refer_lengths (B,).
        Returns content (B, 256, T), prompt (B, Lp, 256), prompt_mask (B, Lp) bool -- what ``Denoiser.sample`` takes.
        ``autocast`` = torch.float16 / torch.bfloat16 runs the GEMMs, convolutions and attention of the two encoders with
        16-bit operands (``torch.autocast``; LayerNorm and the outputs stay fp32) -- the reference's own inference runs fp32."""
        with torch.autocast(c_padded.device.type, dtype=autocast, enabled=autocast is not None):
            g = self.ref_enc(refer_padded.transpose(1, 2)).unsqueeze(-1)               # (B, 100, 1)
            prompt = self.prompt_encoder(refer_padded, refer_lengths)
            content = self.phoneme_encoder(c_padded, lengths, g).transpose(1, 2)
        mask = torch.arange(refer_padded.shape[2], device=refer_padded.device)[None, :] < refer_lengths[:, None]
        return content.float().contiguous(), prompt.float().contiguous(), mask


def pre_model_state_from_checkpoint(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """the ``pre_model.*`` tensors of a full NS2VC checkpoint (``torch.load(path)['model']``), prefix stripped"""
    return {k[len("pre_model."):]: v for k, v in state.items() if k.startswith("pre_model.")}
