"""ORACLE (test infrastructure, never shipped, never the thing measured).

A functional PyTorch-CPU fp32 restatement of the NS2VC denoiser forward,
``UNet1DConditionModel.forward`` (reference ``unet1d/unet_1d_condition.py:743-1037``)
plus the ``Diffusion_Encoder`` adapter (``model.py:403-415``).  It is written
from the op-level description in SURVEY.md Appendix A, operates on a plain
``{name: tensor}`` state dict with the reference's key names, and is pinned to
the imported reference by ``tests/golden/make_golden.py`` (run in the build
container, where ``/root/reference`` exists) — the resulting fixtures under
``tests/golden/`` are what the GPU box checks against.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from ns2vc_amd.spec import UNetConfig, topology


def _t(P: Dict[str, object], name: str) -> torch.Tensor:
    v = P[name]
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(v)
    return v


def as_torch_state(P: Dict[str, object]) -> Dict[str, torch.Tensor]:
    return {k: _t(P, k).float().contiguous() for k in P}


# -- embeddings -------------------------------------------------------------
def sinusoid(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """embeddings.py:24-64 with flip_sin_to_cos=True, freq_shift=0 -> [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = timesteps[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def time_mlp(P, t_emb: torch.Tensor) -> torch.Tensor:
    """embeddings.py:157-201: Linear, SiLU, Linear."""
    h = F.linear(t_emb, _t(P, "time_embedding.linear_1.weight"), _t(P, "time_embedding.linear_1.bias"))
    return F.linear(F.silu(h), _t(P, "time_embedding.linear_2.weight"), _t(P, "time_embedding.linear_2.bias"))


def prompt_pool_embedding(P, cfg: UNetConfig, prompt: torch.Tensor) -> torch.Tensor:
    """TextTimeEmbedding + AttentionPooling (embeddings.py:421-434, 499-546).
    The prompt mask is NOT applied here (reference quirk, kept)."""
    pre = "add_embedding"
    B, L, W = prompt.shape
    H = cfg.addition_embed_heads
    dph = W // H
    x = F.layer_norm(prompt, (W,), _t(P, f"{pre}.norm1.weight"), _t(P, f"{pre}.norm1.bias"), 1e-5)
    cls = x.mean(dim=1, keepdim=True) + _t(P, f"{pre}.pool.positional_embedding")
    seq = torch.cat([cls, x], dim=1)                                   # (B, L+1, W)
    q = F.linear(cls, _t(P, f"{pre}.pool.q_proj.weight"), _t(P, f"{pre}.pool.q_proj.bias"))
    k = F.linear(seq, _t(P, f"{pre}.pool.k_proj.weight"), _t(P, f"{pre}.pool.k_proj.bias"))
    v = F.linear(seq, _t(P, f"{pre}.pool.v_proj.weight"), _t(P, f"{pre}.pool.v_proj.bias"))
    q = q.view(B, 1, H, dph).permute(0, 2, 1, 3)                       # (B,H,1,d)
    k = k.view(B, L + 1, H, dph).permute(0, 2, 1, 3)
    v = v.view(B, L + 1, H, dph).permute(0, 2, 1, 3)
    s = dph ** -0.25
    w = torch.softmax(((q * s) @ (k * s).transpose(-1, -2)).float(), dim=-1)   # (B,H,1,L+1)
    pooled = (w @ v).permute(0, 2, 1, 3).reshape(B, W)                # width index = head*dph + c
    y = F.linear(pooled, _t(P, f"{pre}.proj.weight"), _t(P, f"{pre}.proj.bias"))
    return F.layer_norm(y, (y.shape[-1],), _t(P, f"{pre}.norm2.weight"), _t(P, f"{pre}.norm2.bias"), 1e-5)


# -- blocks -----------------------------------------------------------------
def resnet_block(P, cfg: UNetConfig, pre: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """ResnetBlock2D with scale_shift time conditioning (resnet.py:591-641)."""
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    h = F.silu(F.group_norm(x, G, _t(P, f"{pre}.norm1.weight"), _t(P, f"{pre}.norm1.bias"), eps))
    h = F.conv1d(h, _t(P, f"{pre}.conv1.weight"), _t(P, f"{pre}.conv1.bias"), padding=1)
    ss = F.linear(F.silu(emb), _t(P, f"{pre}.time_emb_proj.weight"), _t(P, f"{pre}.time_emb_proj.bias"))[:, :, None]
    scale, shift = ss.chunk(2, dim=1)
    h = F.group_norm(h, G, _t(P, f"{pre}.norm2.weight"), _t(P, f"{pre}.norm2.bias"), eps) * (1 + scale) + shift
    h = F.conv1d(F.silu(h), _t(P, f"{pre}.conv2.weight"), _t(P, f"{pre}.conv2.bias"), padding=1)
    if f"{pre}.conv_shortcut.weight" in P:
        x = F.conv1d(x, _t(P, f"{pre}.conv_shortcut.weight"), _t(P, f"{pre}.conv_shortcut.bias"))
    return x + h


def _attention(P, pre: str, heads: int, x: torch.Tensor, ctx: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Attention + AttnProcessor2_0 (attention_processor.py:980-1052); bias (B,1,Lk) additive."""
    B, Lq, D = x.shape
    q = F.linear(x, _t(P, f"{pre}.to_q.weight"))
    k = F.linear(ctx, _t(P, f"{pre}.to_k.weight"))
    v = F.linear(ctx, _t(P, f"{pre}.to_v.weight"))
    hd = D // heads
    q = q.view(B, Lq, heads, hd).transpose(1, 2)
    k = k.view(B, -1, heads, hd).transpose(1, 2)
    v = v.view(B, -1, heads, hd).transpose(1, 2)
    m = None if bias is None else bias[:, None, :, :].expand(B, heads, 1, bias.shape[-1])
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=m, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, Lq, D)
    return F.linear(o, _t(P, f"{pre}.to_out.0.weight"), _t(P, f"{pre}.to_out.0.bias"))


def transformer_block(P, cfg: UNetConfig, pre: str, x: torch.Tensor, prompt: torch.Tensor,
                      bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Transformer2DModel continuous branch + one BasicTransformerBlock
    (transformer_1d.py:256-295, attention.py:130-203, GEGLU attention.py:280-301)."""
    D = x.shape[1]
    res = x
    h = F.group_norm(x, cfg.norm_num_groups, _t(P, f"{pre}.norm.weight"), _t(P, f"{pre}.norm.bias"), cfg.attn_norm_eps)
    h = F.conv1d(h, _t(P, f"{pre}.proj_in.weight"), _t(P, f"{pre}.proj_in.bias"))
    y = h.permute(0, 2, 1)                                              # (B,T,D)
    t = f"{pre}.transformer_blocks.0"
    n = F.layer_norm(y, (D,), _t(P, f"{t}.norm1.weight"), _t(P, f"{t}.norm1.bias"), 1e-5)
    y = y + _attention(P, f"{t}.attn1", cfg.heads, n, n, None)
    n = F.layer_norm(y, (D,), _t(P, f"{t}.norm2.weight"), _t(P, f"{t}.norm2.bias"), 1e-5)
    y = y + _attention(P, f"{t}.attn2", cfg.heads, n, prompt, bias)
    n = F.layer_norm(y, (D,), _t(P, f"{t}.norm3.weight"), _t(P, f"{t}.norm3.bias"), 1e-5)
    u = F.linear(n, _t(P, f"{t}.ff.net.0.proj.weight"), _t(P, f"{t}.ff.net.0.proj.bias"))
    a, g = u.chunk(2, dim=-1)
    y = y + F.linear(a * F.gelu(g), _t(P, f"{t}.ff.net.2.weight"), _t(P, f"{t}.ff.net.2.bias"))
    h = F.conv1d(y.permute(0, 2, 1), _t(P, f"{pre}.proj_out.weight"), _t(P, f"{pre}.proj_out.bias"))
    return h + res


# -- whole forward ------------------------------------------------------------
@torch.no_grad()
def unet_forward(P: Dict[str, object], cfg: UNetConfig, sample: torch.Tensor, timestep, prompt: torch.Tensor,
                 prompt_mask: Optional[torch.Tensor] = None, taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """sample (B, in_channels, T) NCT; timestep (B,) / 0-d / python scalar;
    prompt (B, Lp, cross); prompt_mask (B, Lp) bool, True=keep.  Returns (B, out, T).
    ``taps`` (optional dict) receives named intermediates for engine debugging."""
    def tap(name, v):
        if taps is not None:
            taps[name] = v.clone()

    B = sample.shape[0]
    bias = None
    if prompt_mask is not None:
        bias = ((1 - prompt_mask.to(sample.dtype)) * -10000.0)[:, None, :]       # (B,1,Lp)
    ts = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
    if ts.ndim == 0:
        ts = ts[None]
    ts = ts.expand(B)
    emb = time_mlp(P, sinusoid(ts, cfg.time_dim)) + prompt_pool_embedding(P, cfg, prompt)
    tap("emb", emb)

    x = F.conv1d(sample, _t(P, "conv_in.weight"), _t(P, "conv_in.bias"), padding=1)
    tap("conv_in", x)
    skips = [x]
    blocks = topology(cfg)
    for b in blocks:
        if b.kind == "up":
            break
        tag = "mid" if b.kind == "mid" else f"down{b.index}"
        if b.kind == "down":
            for j, r in enumerate(b.resnets):
                x = resnet_block(P, cfg, r.prefix, x, emb)
                tap(f"{tag}.res{j}", x)
                if b.attns:
                    x = transformer_block(P, cfg, b.attns[j].prefix, x, prompt, bias)
                    tap(f"{tag}.attn{j}", x)
                skips.append(x)
            if b.sampler:
                x = F.conv1d(x, _t(P, f"{b.sampler_prefix}.weight"), _t(P, f"{b.sampler_prefix}.bias"), stride=2, padding=1)
                tap(f"{tag}.ds", x)
                skips.append(x)
        else:  # mid: resnet, attn, resnet (unet_1d_blocks.py:602-623)
            x = resnet_block(P, cfg, b.resnets[0].prefix, x, emb)
            tap("mid.res0", x)
            x = transformer_block(P, cfg, b.attns[0].prefix, x, prompt, bias)
            tap("mid.attn0", x)
            x = resnet_block(P, cfg, b.resnets[1].prefix, x, emb)
            tap("mid.res1", x)
    for b in blocks:
        if b.kind != "up":
            continue
        tag = f"up{b.index}"
        for j, r in enumerate(b.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(P, cfg, r.prefix, x, emb)
            tap(f"{tag}.res{j}", x)
            if b.attns:
                x = transformer_block(P, cfg, b.attns[j].prefix, x, prompt, bias)
                tap(f"{tag}.attn{j}", x)
        if b.sampler:
            # explicit-size nearest upsample to the next skip's length
            x = F.interpolate(x, size=skips[-1].shape[-1], mode="nearest")
            x = F.conv1d(x, _t(P, f"{b.sampler_prefix}.weight"), _t(P, f"{b.sampler_prefix}.bias"), padding=1)
            tap(f"{tag}.us", x)
    x = F.silu(F.group_norm(x, cfg.norm_num_groups, _t(P, "conv_norm_out.weight"), _t(P, "conv_norm_out.bias"), cfg.norm_eps))
    x = F.conv1d(x, _t(P, "conv_out.weight"), _t(P, "conv_out.bias"), padding=1)
    tap("out", x)
    return x


@torch.no_grad()
def denoiser(P, cfg: UNetConfig, x: torch.Tensor, content: torch.Tensor, prompt: torch.Tensor,
             prompt_mask: Optional[torch.Tensor], t: torch.Tensor) -> torch.Tensor:
    """Diffusion_Encoder.forward (model.py:403-415) on batch-first tensors:
    x (B,100,T), content (B,256,T), prompt (B,Lp,256), mask (B,Lp) -> x0_pred (B,100,T)."""
    return unet_forward(P, cfg, torch.cat([x, content], dim=1), t, prompt, prompt_mask)
