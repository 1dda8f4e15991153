"""Differentiable PyTorch forward of the denoiser, for TRAINING only.

The HIP engine is inference-only.  The reference trains through the same module
(``model.py:720`` calls the UNet under autograd; ``train.py`` -> ``Trainer.train``,
``model.py:839-946``), so that ``train.py`` keeps working against this package the
drop-in ``UNet1DConditionModel`` routes a call here -- and ONLY here -- when autograd
is recording (``torch.is_grad_enabled()`` and a parameter or the input requires grad).
Inference (``torch.no_grad()``, as ``NaturalSpeech2.sample`` ``model.py:605`` and
``Svc.infer`` ``infer_tool.py:199`` run it) never takes this path: it goes to the HIP
engine or raises.  The ops below are ordinary PyTorch(-ROCm) autograd ops on the
module's own parameters; nothing here is used by the benchmarks or the parity tests
of the engine, and nothing imports ``oracle/``.

Math restated from SURVEY.md Appendix A (reference ``unet1d/unet_1d_condition.py:743-1037``,
``resnet.py:591-641``, ``transformer_1d.py:256-295``, ``attention.py:130-203,280-301``,
``attention_processor.py:980-1052``, ``embeddings.py:24-64,157-201,421-434,499-546``).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from ns2vc_amd.spec import UNetConfig, topology


class TorchDenoiser:
    """Stateless evaluator bound to an ``nn.Module`` whose parameters carry the reference's names."""

    def __init__(self, module: torch.nn.Module, cfg: UNetConfig):
        self.m, self.cfg = module, cfg
        self.blocks = topology(cfg)

    # ---- parameter access by reference name -------------------------------------------------------------------------
    def p(self, name: str) -> torch.Tensor:
        return self.m.get_parameter(name)

    def has(self, name: str) -> bool:
        try:
            self.m.get_parameter(name)
            return True
        except AttributeError:
            return False

    def lin(self, x, pre: str, bias: bool = True):
        return F.linear(x, self.p(pre + ".weight"), self.p(pre + ".bias") if bias else None)

    def conv(self, x, pre: str, **kw):
        return F.conv1d(x, self.p(pre + ".weight"), self.p(pre + ".bias"), **kw)

    def gn(self, x, pre: str, eps: float):
        return F.group_norm(x, self.cfg.norm_num_groups, self.p(pre + ".weight"), self.p(pre + ".bias"), eps)

    def ln(self, x, pre: str):
        return F.layer_norm(x, (x.shape[-1],), self.p(pre + ".weight"), self.p(pre + ".bias"), 1e-5)

    # ---- embeddings -------------------------------------------------------------------------------------------------
    def time_embedding(self, t: torch.Tensor) -> torch.Tensor:
        half = self.cfg.time_dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        ang = t[:, None].float() * freqs[None, :]
        e = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)            # flip_sin_to_cos=True, freq_shift=0
        return self.lin(F.silu(self.lin(e, "time_embedding.linear_1")), "time_embedding.linear_2")

    def prompt_embedding(self, prompt: torch.Tensor) -> torch.Tensor:
        """TextTimeEmbedding / AttentionPooling; the prompt mask is NOT applied (reference behaviour)."""
        B, L, W = prompt.shape
        H = self.cfg.addition_embed_heads
        d = W // H
        x = self.ln(prompt, "add_embedding.norm1")
        cls = x.mean(dim=1, keepdim=True) + self.p("add_embedding.pool.positional_embedding")
        seq = torch.cat([cls, x], dim=1)
        q = self.lin(cls, "add_embedding.pool.q_proj").view(B, 1, H, d).transpose(1, 2)
        k = self.lin(seq, "add_embedding.pool.k_proj").view(B, L + 1, H, d).transpose(1, 2)
        v = self.lin(seq, "add_embedding.pool.v_proj").view(B, L + 1, H, d).transpose(1, 2)
        w = torch.softmax((q @ k.transpose(-1, -2)).float() / math.sqrt(d), dim=-1).to(v.dtype)
        pooled = (w @ v).transpose(1, 2).reshape(B, W)
        return self.ln(self.lin(pooled, "add_embedding.proj"), "add_embedding.norm2")

    # ---- blocks -----------------------------------------------------------------------------------------------------
    def resnet(self, pre: str, x, emb):
        h = self.conv(F.silu(self.gn(x, pre + ".norm1", self.cfg.norm_eps)), pre + ".conv1", padding=1)
        scale, shift = self.lin(F.silu(emb), pre + ".time_emb_proj")[:, :, None].chunk(2, dim=1)
        h = self.gn(h, pre + ".norm2", self.cfg.norm_eps) * (1 + scale) + shift
        h = self.conv(F.silu(h), pre + ".conv2", padding=1)
        if self.has(pre + ".conv_shortcut.weight"):
            x = self.conv(x, pre + ".conv_shortcut")
        return x + h

    def attention(self, pre: str, x, ctx, bias):
        B, Lq, D = x.shape
        H = self.cfg.heads
        q = self.lin(x, pre + ".to_q", bias=False).view(B, Lq, H, D // H).transpose(1, 2)
        k = self.lin(ctx, pre + ".to_k", bias=False).view(B, -1, H, D // H).transpose(1, 2)
        v = self.lin(ctx, pre + ".to_v", bias=False).view(B, -1, H, D // H).transpose(1, 2)
        # explicit softmax(q k^T / sqrt(d) + bias) v: the fused fp32 SDPA kernel of PyTorch-ROCm on gfx950 is 3.4e-3 off the
        # reference golden on this model (tools/torch_fp32_probe.py; its MATH backend is 9e-7), and this path is about
        # exact training numerics, not speed
        s = (q @ k.transpose(-1, -2)) * (D // H) ** -0.5
        if bias is not None:
            s = s + bias[:, None, :, :]
        o = torch.softmax(s, dim=-1) @ v
        return self.lin(o.transpose(1, 2).reshape(B, Lq, D), pre + ".to_out.0")

    def transformer(self, pre: str, x, prompt, bias):
        t = pre + ".transformer_blocks.0"
        y = self.conv(self.gn(x, pre + ".norm", self.cfg.attn_norm_eps), pre + ".proj_in").permute(0, 2, 1)
        n = self.ln(y, t + ".norm1")
        y = y + self.attention(t + ".attn1", n, n, None)
        y = y + self.attention(t + ".attn2", self.ln(y, t + ".norm2"), prompt, bias)
        a, g = self.lin(self.ln(y, t + ".norm3"), t + ".ff.net.0.proj").chunk(2, dim=-1)
        y = y + self.lin(a * F.gelu(g), t + ".ff.net.2")
        return self.conv(y.permute(0, 2, 1), pre + ".proj_out") + x

    # ---- whole forward ----------------------------------------------------------------------------------------------
    def __call__(self, sample: torch.Tensor, timestep: torch.Tensor, prompt: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
        bias = None if mask is None else ((1 - mask.to(sample.dtype)) * -10000.0)[:, None, :]
        emb = self.time_embedding(timestep) + self.prompt_embedding(prompt)
        x = self.conv(sample, "conv_in", padding=1)
        skips = [x]
        for b in self.blocks:
            if b.kind == "down":
                for j, r in enumerate(b.resnets):
                    x = self.resnet(r.prefix, x, emb)
                    if b.attns:
                        x = self.transformer(b.attns[j].prefix, x, prompt, bias)
                    skips.append(x)
                if b.sampler:
                    x = self.conv(x, b.sampler_prefix, stride=2, padding=1)
                    skips.append(x)
            elif b.kind == "mid":
                x = self.resnet(b.resnets[0].prefix, x, emb)
                x = self.transformer(b.attns[0].prefix, x, prompt, bias)
                x = self.resnet(b.resnets[1].prefix, x, emb)
            else:
                for j, r in enumerate(b.resnets):
                    x = self.resnet(r.prefix, torch.cat([x, skips.pop()], dim=1), emb)
                    if b.attns:
                        x = self.transformer(b.attns[j].prefix, x, prompt, bias)
                if b.sampler:      # explicit-size nearest upsample to the next skip's length, then conv
                    x = self.conv(F.interpolate(x, size=skips[-1].shape[-1], mode="nearest"), b.sampler_prefix, padding=1)
        x = F.silu(self.gn(x, "conv_norm_out", self.cfg.norm_eps))
        return self.conv(x, "conv_out", padding=1)
