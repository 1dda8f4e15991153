"""Denoiser topology and parameter specification.

The engine, the drop-in ``unet1d`` module, the procedural weight generator and
the tests all derive the list of parameter tensors from one place: this file.
It restates *which tensors exist and what shape they have* for the reference
``UNet1DConditionModel`` as configured by NS2VC (reference
``model.py:391-400`` ctor kwargs, defaults ``unet1d/unet_1d_condition.py:151-203``,
block wiring ``unet1d/unet_1d_blocks.py:861,1019,516,1986,2134``).

Nothing here touches a GPU.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    """Subset of the reference ctor kwargs that NS2VC actually varies.

    ``attention_head_dim`` keeps the reference's (diffusers) naming wart: it is
    the NUMBER OF HEADS (``unet_1d_condition.py:213-219``); the per-head width
    at level l is ``block_out_channels[l] // attention_head_dim``.
    """

    in_channels: int = 356           # latent (100) + content (256), model.py:392
    out_channels: int = 100
    block_out_channels: Tuple[int, ...] = (128, 256, 384, 512)
    norm_num_groups: int = 8
    cross_attention_dim: int = 256
    attention_head_dim: int = 8      # = number of heads
    layers_per_block: int = 2
    addition_embed_type: str = "text"
    resnet_time_scale_shift: str = "scale_shift"
    # fixed by the reference defaults, listed for the record
    norm_eps: float = 1e-5           # resnet / conv_norm_out GroupNorm eps
    attn_norm_eps: float = 1e-6      # Transformer2DModel GroupNorm eps (transformer_1d.py:134)
    addition_embed_heads: int = 64   # unet_1d_condition.py:173 default
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3

    @property
    def latent_channels(self) -> int:
        return self.out_channels

    @property
    def content_channels(self) -> int:
        return self.in_channels - self.out_channels

    @property
    def heads(self) -> int:
        return self.attention_head_dim

    @property
    def time_dim(self) -> int:       # sinusoid width
        return self.block_out_channels[0]

    @property
    def temb_dim(self) -> int:       # time embedding width
        return self.block_out_channels[0] * 4

    def validate(self) -> None:
        n = len(self.block_out_channels)
        if len(self.down_block_types) != n or len(self.up_block_types) != n:
            raise ValueError("block type tuples must match block_out_channels in length")
        if self.addition_embed_type != "text":
            raise ValueError("only addition_embed_type='text' is supported (NS2VC config)")
        if self.resnet_time_scale_shift != "scale_shift":
            raise ValueError("only resnet_time_scale_shift='scale_shift' is supported (NS2VC config)")
        for c in self.block_out_channels:
            if c % self.norm_num_groups or c % self.heads:
                raise ValueError(f"channels {c} not divisible by groups/heads")


# ----------------------------------------------------------------------------
# Topology: a flat description of the blocks, shared by spec / oracle / engine
# ----------------------------------------------------------------------------
@dataclass
class ResnetSpec:
    prefix: str
    cin: int
    cout: int

    @property
    def has_shortcut(self) -> bool:
        return self.cin != self.cout


@dataclass
class AttnSpec:
    prefix: str
    dim: int


@dataclass
class BlockSpec:
    kind: str                     # 'down' | 'mid' | 'up'
    index: int
    level: int                    # resolution level of the block's resnets
    resnets: List[ResnetSpec] = field(default_factory=list)
    attns: List[AttnSpec] = field(default_factory=list)
    sampler: str | None = None    # 'down' | 'up' | None
    sampler_prefix: str | None = None
    channels: int = 0


def topology(cfg: UNetConfig) -> List[BlockSpec]:
    """Block list in execution order (down*, mid, up*)."""
    cfg.validate()
    chans = cfg.block_out_channels
    n = len(chans)
    blocks: List[BlockSpec] = []
    out_c = chans[0]
    for i, typ in enumerate(cfg.down_block_types):
        in_c, out_c = out_c, chans[i]
        b = BlockSpec("down", i, i, channels=out_c)
        for j in range(cfg.layers_per_block):
            b.resnets.append(ResnetSpec(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c))
            if typ == "CrossAttnDownBlock2D":
                b.attns.append(AttnSpec(f"down_blocks.{i}.attentions.{j}", out_c))
        if i != n - 1:
            b.sampler, b.sampler_prefix = "down", f"down_blocks.{i}.downsamplers.0.conv"
        blocks.append(b)
    mid_c = chans[-1]
    m = BlockSpec("mid", 0, n - 1, channels=mid_c)
    m.resnets = [ResnetSpec("mid_block.resnets.0", mid_c, mid_c), ResnetSpec("mid_block.resnets.1", mid_c, mid_c)]
    m.attns = [AttnSpec("mid_block.attentions.0", mid_c)]
    blocks.append(m)
    rev = list(reversed(chans))
    out_c = rev[0]
    for i, typ in enumerate(cfg.up_block_types):
        prev_c, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, n - 1)]
        b = BlockSpec("up", i, n - 1 - i, channels=out_c)
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            skip_c = in_c if j == nl - 1 else out_c
            res_in = prev_c if j == 0 else out_c
            b.resnets.append(ResnetSpec(f"up_blocks.{i}.resnets.{j}", res_in + skip_c, out_c))
            if typ == "CrossAttnUpBlock2D":
                b.attns.append(AttnSpec(f"up_blocks.{i}.attentions.{j}", out_c))
        if i != n - 1:
            b.sampler, b.sampler_prefix = "up", f"up_blocks.{i}.upsamplers.0.conv"
        blocks.append(b)
    return blocks


def attention_workgroups_per_forward(cfg: UNetConfig, B: int, T: int) -> int:
    """Workgroups of all attention launches of one denoiser evaluation (a workgroup = 128 queries of one head of one batch item; self- and
    cross-attention of every transformer block): the denominator of the optimistic-pass fallback rate (``Engine.attn_fallbacks``)."""
    n = 0
    for b in topology(cfg):
        Tl = T
        for _ in range(b.level):
            Tl = (Tl + 1) // 2
        n += len(b.attns) * 2 * ((Tl + 127) // 128) * cfg.heads * B
    return n


def _resnet_params(r: ResnetSpec, temb: int) -> List[Tuple[str, Tuple[int, ...]]]:
    p = [
        (f"{r.prefix}.norm1.weight", (r.cin,)), (f"{r.prefix}.norm1.bias", (r.cin,)),
        (f"{r.prefix}.conv1.weight", (r.cout, r.cin, 3)), (f"{r.prefix}.conv1.bias", (r.cout,)),
        (f"{r.prefix}.time_emb_proj.weight", (2 * r.cout, temb)), (f"{r.prefix}.time_emb_proj.bias", (2 * r.cout,)),
        (f"{r.prefix}.norm2.weight", (r.cout,)), (f"{r.prefix}.norm2.bias", (r.cout,)),
        (f"{r.prefix}.conv2.weight", (r.cout, r.cout, 3)), (f"{r.prefix}.conv2.bias", (r.cout,)),
    ]
    if r.has_shortcut:
        p += [(f"{r.prefix}.conv_shortcut.weight", (r.cout, r.cin, 1)), (f"{r.prefix}.conv_shortcut.bias", (r.cout,))]
    return p


def _attn_params(a: AttnSpec, cross: int) -> List[Tuple[str, Tuple[int, ...]]]:
    d, t = a.dim, f"{a.prefix}.transformer_blocks.0"
    return [
        (f"{a.prefix}.norm.weight", (d,)), (f"{a.prefix}.norm.bias", (d,)),
        (f"{a.prefix}.proj_in.weight", (d, d, 1)), (f"{a.prefix}.proj_in.bias", (d,)),
        (f"{t}.norm1.weight", (d,)), (f"{t}.norm1.bias", (d,)),
        (f"{t}.attn1.to_q.weight", (d, d)), (f"{t}.attn1.to_k.weight", (d, d)), (f"{t}.attn1.to_v.weight", (d, d)),
        (f"{t}.attn1.to_out.0.weight", (d, d)), (f"{t}.attn1.to_out.0.bias", (d,)),
        (f"{t}.norm2.weight", (d,)), (f"{t}.norm2.bias", (d,)),
        (f"{t}.attn2.to_q.weight", (d, d)), (f"{t}.attn2.to_k.weight", (d, cross)), (f"{t}.attn2.to_v.weight", (d, cross)),
        (f"{t}.attn2.to_out.0.weight", (d, d)), (f"{t}.attn2.to_out.0.bias", (d,)),
        (f"{t}.norm3.weight", (d,)), (f"{t}.norm3.bias", (d,)),
        (f"{t}.ff.net.0.proj.weight", (8 * d, d)), (f"{t}.ff.net.0.proj.bias", (8 * d,)),
        (f"{t}.ff.net.2.weight", (d, 4 * d)), (f"{t}.ff.net.2.bias", (d,)),
        (f"{a.prefix}.proj_out.weight", (d, d, 1)), (f"{a.prefix}.proj_out.bias", (d,)),
    ]


def param_spec(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for every tensor in the reference state_dict, in the
    reference's registration order (attentions before resnets inside a block)."""
    c0, temb, cross = cfg.block_out_channels[0], cfg.temb_dim, cfg.cross_attention_dim
    items: List[Tuple[str, Tuple[int, ...]]] = [
        ("conv_in.weight", (c0, cfg.in_channels, 3)), ("conv_in.bias", (c0,)),
        ("time_embedding.linear_1.weight", (temb, cfg.time_dim)), ("time_embedding.linear_1.bias", (temb,)),
        ("time_embedding.linear_2.weight", (temb, temb)), ("time_embedding.linear_2.bias", (temb,)),
        ("add_embedding.norm1.weight", (cross,)), ("add_embedding.norm1.bias", (cross,)),
        ("add_embedding.pool.positional_embedding", (1, cross)),
        ("add_embedding.pool.k_proj.weight", (cross, cross)), ("add_embedding.pool.k_proj.bias", (cross,)),
        ("add_embedding.pool.q_proj.weight", (cross, cross)), ("add_embedding.pool.q_proj.bias", (cross,)),
        ("add_embedding.pool.v_proj.weight", (cross, cross)), ("add_embedding.pool.v_proj.bias", (cross,)),
        ("add_embedding.proj.weight", (temb, cross)), ("add_embedding.proj.bias", (temb,)),
        ("add_embedding.norm2.weight", (temb,)), ("add_embedding.norm2.bias", (temb,)),
    ]
    topo = topology(cfg)
    # the reference registers up_blocks before mid_block (unet_1d_condition.py ctor)
    for b in [b for b in topo if b.kind == "down"] + [b for b in topo if b.kind == "up"] + [b for b in topo if b.kind == "mid"]:
        for a in b.attns:
            items += _attn_params(a, cross)
        for r in b.resnets:
            items += _resnet_params(r, temb)
        if b.sampler:
            items += [(f"{b.sampler_prefix}.weight", (b.channels, b.channels, 3)), (f"{b.sampler_prefix}.bias", (b.channels,))]
    items += [
        ("conv_norm_out.weight", (c0,)), ("conv_norm_out.bias", (c0,)),
        ("conv_out.weight", (cfg.out_channels, c0, 3)), ("conv_out.bias", (cfg.out_channels,)),
    ]
    spec = OrderedDict(items)
    assert len(spec) == len(items), "duplicate parameter name"
    return spec


def level_lengths(T: int, n_levels: int) -> List[int]:
    """Frame count per resolution level: stride-2, k3, pad1 conv => ceil(T/2)
    (reference resnet.py:176-223)."""
    out = [T]
    for _ in range(n_levels - 1):
        out.append((out[-1] + 1) // 2)
    return out


def frames_for_seconds(seconds: float, sr: int = 24000, hop: int = 256) -> int:
    """Vocos mel frames for an utterance: floor(sr*s/hop)+1."""
    return int(sr * seconds) // hop + 1


def algorithmic_gflop_per_sample_step(T: int, Lp: int) -> float:
    """Reference-math FLOPs (2*MAC) of ONE UNet forward for one sample.

    Published values: (188,469)=9.027, (938,469)=39.157,
    (2813,469)=135.834, (2813,1407)=159.520 GFLOP.  bench.py uses the published
    table when the shape is in it and this analytic count otherwise; the count
    below reproduces the table to <0.1 %.
    """
    cfg = UNetConfig()
    Ts = level_lengths(T, len(cfg.block_out_channels))
    fl = 0.0
    c0, temb, cross = cfg.block_out_channels[0], cfg.temb_dim, cfg.cross_attention_dim
    fl += 2.0 * T * c0 * cfg.in_channels * 3                      # conv_in
    fl += 2.0 * (cfg.time_dim * temb + temb * temb)               # time mlp
    # add_embedding: k/v proj over Lp+1 tokens, q on 1, pooling, proj
    fl += 2.0 * (2 * (Lp + 1) * cross * cross + cross * cross + cross * temb) + 4.0 * (Lp + 1) * cross
    for b in topology(cfg):
        Tl = Ts[b.level]
        for r in b.resnets:
            fl += 2.0 * Tl * r.cout * r.cin * 3 + 2.0 * Tl * r.cout * r.cout * 3
            fl += 2.0 * temb * 2 * r.cout
            if r.has_shortcut:
                fl += 2.0 * Tl * r.cout * r.cin
        for a in b.attns:
            d = a.dim
            fl += 2.0 * Tl * d * d * 2                              # proj_in/out
            fl += 2.0 * Tl * d * d * 4 + 2.0 * Tl * d * d * 2       # self q,k,v,o + cross q,o
            fl += 2.0 * Lp * cross * d * 2                          # cross k,v
            fl += 2.0 * Tl * d * 8 * d + 2.0 * Tl * 4 * d * d       # ff
            fl += 4.0 * Tl * Tl * d + 4.0 * Tl * Lp * d             # sdpa self + cross
        if b.sampler == "down":
            fl += 2.0 * Ts[b.level + 1] * b.channels * b.channels * 3
        elif b.sampler == "up":
            fl += 2.0 * Ts[b.level - 1] * b.channels * b.channels * 3
    fl += 2.0 * T * cfg.out_channels * c0 * 3
    return fl / 1e9


PUBLISHED_GFLOP = {(188, 469): 9.027, (938, 469): 39.157, (2813, 469): 135.834, (2813, 1407): 159.520}
