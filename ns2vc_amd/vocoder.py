"""Vocos (mel -> waveform) back end on PyTorch-ROCm -- SURVEY 8(f) rank 2, BASELINE north_star "Vocos mel/iSTFT stay on
PyTorch-ROCm and stream-overlap the denoiser".

The reference decodes the sampled latent with ``vocos.decode(latent)`` (``model.py:689-691``; the latent IS a 100-bin
log-mel at 24 kHz / hop 256), where ``vocos`` is ``Vocos.from_pretrained("charactr/vocos-mel-24khz")`` (``model.py:762``,
``inference/infer_tool.py:135``) -- a pip dependency that is not vendored in the reference tree and not installed in this
image.  This module restates the published architecture of that checkpoint so that the back end is a REAL stage of the
pipeline (``OverlappedPipeline(post_fn=VocosDecoder.decode)``) instead of a hook:

  backbone  Conv1d(100 -> 512, k 7) -> LayerNorm(eps 1e-6) -> 8 x ConvNeXt block [depthwise Conv1d k 7 -> LayerNorm ->
            Linear 512 -> 1536 -> GELU -> Linear 1536 -> 512 -> per-channel gamma -> + residual] -> LayerNorm
  head      Linear 512 -> 1026 = (log-magnitude | phase) of a 513-bin spectrum; mag = min(exp(.), 100);
            S = mag * (cos p + i sin p); inverse STFT, n_fft 1024, hop 256, Hann window, "center" padding

with the parameter names of the ``vocos`` package (``backbone.embed``, ``backbone.norm``, ``backbone.convnext.N.{dwconv,norm,
pwconv1,pwconv2,gamma}``, ``backbone.final_layer_norm``, ``head.out``, ``head.istft.window``), so ``load_state_dict`` takes a
``pytorch_model.bin`` of ``charactr/vocos-mel-24khz`` (its ``feature_extractor.*`` buffers are ignored: decoding does not use them).

**Parity unpinned**: neither the ``vocos`` package nor its checkpoint is available offline, so this restatement has no
golden vector of the reference's back end behind it; the tests check the inverse STFT against an independent numpy
overlap-add and the module's shapes / key names only.  End-to-end RTF figures that include it say so.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim: int, intermediate_dim: int, layer_scale_init_value: float):
        super().__init__()
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, intermediate_dim)
        self.pwconv2 = nn.Linear(intermediate_dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:          # (B, C, T)
        y = self.dwconv(x).transpose(1, 2)
        y = self.pwconv2(F.gelu(self.pwconv1(self.norm(y))))
        return x + (self.gamma * y).transpose(1, 2)


class VocosBackbone(nn.Module):
    def __init__(self, input_channels: int = 100, dim: int = 512, intermediate_dim: int = 1536, num_layers: int = 8):
        super().__init__()
        self.embed = nn.Conv1d(input_channels, dim, kernel_size=7, padding=3)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.convnext = nn.ModuleList([ConvNeXtBlock(dim, intermediate_dim, 1.0 / num_layers) for _ in range(num_layers)])
        self.final_layer_norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, mel: torch.Tensor) -> torch.Tensor:        # (B, 100, T) -> (B, T, dim)
        x = self.embed(mel)
        x = self.norm(x.transpose(1, 2)).transpose(1, 2)
        for blk in self.convnext:
            x = blk(x)
        return self.final_layer_norm(x.transpose(1, 2))


class ISTFT(nn.Module):
    """inverse STFT with "center" padding (torch.istft), window kept as a buffer named like the package's"""

    def __init__(self, n_fft: int, hop_length: int, win_length: int):
        super().__init__()
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.register_buffer("window", torch.hann_window(win_length))

    def forward(self, spec: torch.Tensor) -> torch.Tensor:       # complex (B, n_fft/2+1, T) -> (B, (T-1)*hop)
        return torch.istft(spec, self.n_fft, self.hop_length, self.win_length, self.window.to(spec.real.dtype), center=True)


class ISTFTHead(nn.Module):
    def __init__(self, dim: int = 512, n_fft: int = 1024, hop_length: int = 256):
        super().__init__()
        self.out = nn.Linear(dim, n_fft + 2)
        self.istft = ISTFT(n_fft, hop_length, n_fft)

    def forward(self, x: torch.Tensor) -> torch.Tensor:          # (B, T, dim) -> (B, samples)
        mag, phase = self.out(x).transpose(1, 2).float().chunk(2, dim=1)
        mag = torch.clip(torch.exp(mag), max=1e2)                # the package's safeguard against exploding magnitudes
        return self.istft(torch.complex(mag * torch.cos(phase), mag * torch.sin(phase)))


class VocosDecoder(nn.Module):
    """``decode(mel (B, 100, T)) -> audio (B, (T - 1) * 256)``: what the reference calls as ``vocos.decode``"""

    def __init__(self, input_channels: int = 100, dim: int = 512, intermediate_dim: int = 1536, num_layers: int = 8,
                 n_fft: int = 1024, hop_length: int = 256):
        super().__init__()
        self.backbone = VocosBackbone(input_channels, dim, intermediate_dim, num_layers)
        self.head = ISTFTHead(dim, n_fft, hop_length)

    def load_vocos_state_dict(self, state: Dict[str, torch.Tensor]) -> None:
        """a ``charactr/vocos-mel-24khz`` state dict: everything under ``backbone.`` / ``head.`` must match, the feature
        extractor's buffers (mel filterbank, STFT window of the ENCODER side) are not part of decoding"""
        own = {k: v for k, v in state.items() if k.startswith(("backbone.", "head."))}
        self.load_state_dict(own, strict=True)

    @torch.no_grad()
    def decode(self, mel: torch.Tensor, autocast=None) -> torch.Tensor:
        """``autocast``: torch.float16 / torch.bfloat16 runs the backbone's convolutions / linears on 16-bit MFMA operands
        (the spectrum head and the inverse STFT stay fp32)"""
        if autocast is not None and mel.is_cuda:
            with torch.autocast("cuda", dtype=autocast):
                h = self.backbone(mel.float())
            return self.head(h.float())
        return self.head(self.backbone(mel.float()))

    forward = decode
