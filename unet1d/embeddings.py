"""``TextTimeEmbedding`` for the conditioning encoder (reference ``model.py:6`` imports
it for ``Pre_model.ref_enc``; reference implementation ``unet1d/embeddings.py:421-434,
499-546``).  This module is OUTSIDE the denoiser hot path (it runs once per utterance
in the PyTorch-ROCm conditioning stage), so it is plain PyTorch;
the denoiser's own add_embedding runs inside the HIP engine.  Parameter names match
the reference so checkpoints load unchanged."""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class AttentionPooling(nn.Module):
    def __init__(self, num_heads: int, embed_dim: int, dtype=None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim, dtype=dtype)
        self.q_proj = nn.Linear(embed_dim, embed_dim, dtype=dtype)
        self.v_proj = nn.Linear(embed_dim, embed_dim, dtype=dtype)
        self.num_heads = num_heads
        self.dim_per_head = embed_dim // num_heads

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, L, W = x.shape
        H, d = self.num_heads, self.dim_per_head
        cls = x.mean(dim=1, keepdim=True) + self.positional_embedding.to(x.dtype)
        seq = torch.cat([cls, x], dim=1)
        q = self.q_proj(cls).view(B, 1, H, d).transpose(1, 2)
        k = self.k_proj(seq).view(B, L + 1, H, d).transpose(1, 2)
        v = self.v_proj(seq).view(B, L + 1, H, d).transpose(1, 2)
        # softmax(q k^T / sqrt(d)) v with the single class-token query
        w = torch.softmax((q @ k.transpose(-1, -2)).float() / d ** 0.5, dim=-1).to(v.dtype)
        return (w @ v).transpose(1, 2).reshape(B, W)


class TextTimeEmbedding(nn.Module):
    def __init__(self, encoder_dim: int, time_embed_dim: int, num_heads: int = 64):
        super().__init__()
        self.norm1 = nn.LayerNorm(encoder_dim)
        self.pool = AttentionPooling(num_heads, encoder_dim)
        self.proj = nn.Linear(encoder_dim, time_embed_dim)
        self.norm2 = nn.LayerNorm(time_embed_dim)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.norm2(self.proj(self.pool(self.norm1(hidden_states))))
