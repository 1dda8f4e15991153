"""ContentVec / HuBERT-base content encoder on PyTorch-ROCm -- SURVEY 8(f) rank 4.

The reference turns the 16 kHz source waveform into the 256-d, 50 Hz "content" tensor with a fairseq model
(``utils.py:209-236``: ``checkpoint_utils.load_model_ensemble_and_task(["hubert/checkpoint_best_legacy_500.pt"])``, then
``extract_features(source, padding_mask, output_layer=12)`` followed by ``final_proj``; called from
``inference/infer_tool.py:163-166``, stretched to the latent frame rate by ``repeat_expand_2d``).  fairseq and the checkpoint
are pip / download dependencies that are neither vendored in the reference tree nor present in this image, so -- like the
Vocos back end (``vocoder.py``) -- this module restates the PUBLISHED architecture of that checkpoint (HuBERT base,
``HubertModel`` of fairseq with the ContentVec ``final_proj``) with fairseq's parameter names, so that the ``model`` entry of
``checkpoint_best_legacy_500.pt`` loads with ``load_fairseq_state_dict``:

  feature_extractor   7 x Conv1d(no bias) 1 -> 512, kernels (10,3,3,3,3,2,2), strides (5,2,2,2,2,2,2) = hop 320, receptive field 400;
                      GroupNorm(512, 512) after the first conv only ("default" mode), GELU after every conv
  layer_norm          LayerNorm(512) on the frames, post_extract_proj Linear 512 -> 768
  encoder.pos_conv    Conv1d(768, 768, k 128, padding 64, groups 16) with weight normalisation over dim 2 (weight_g, weight_v),
                      the last frame dropped (even kernel), GELU; added to the frames, then encoder.layer_norm (post-LN encoder)
  encoder.layers.N    12 x [MultiheadAttention(768, 12 heads) -> + residual -> self_attn_layer_norm -> fc1 768 -> 3072 -> GELU ->
                      fc2 -> + residual -> final_layer_norm]
  final_proj          Linear 768 -> 256  (what ContentVec trains; the reference applies it to layer 12's output)

``ContentVec.extract(wav16k)`` = the reference's ``get_hubert_content``: (B, samples) -> (B, 256, frames), frames =
floor((samples - 400) / 320) + 1.  No padding mask (the reference passes an all-False one), no dropout / layer drop (eval).

**Parity unpinned**: neither fairseq nor the checkpoint is available offline; the tests check the frame count, the
parameter names / shapes of the published model (94.57 M parameters with ``final_proj``), the weight-norm reconstruction and
the attention against ``torch.nn.functional.multi_head_attention_forward`` -- not the reference's outputs.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn

CONV_LAYERS = ((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2       # (channels, kernel, stride): fairseq's conv_feature_layers default


class _ConvBlock(nn.Sequential):
    """fairseq names the parts of a feature-extractor block by position: 0 = conv, 2 = GroupNorm (first block only)"""

    def __init__(self, cin: int, cout: int, k: int, stride: int, group_norm: bool):
        layers = [nn.Conv1d(cin, cout, k, stride=stride, bias=False), nn.Dropout(0.0)]
        if group_norm:
            layers.append(nn.GroupNorm(cout, cout, affine=True))
        layers.append(nn.GELU())
        super().__init__(*layers)


class ConvFeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        blocks, cin = [], 1
        for i, (c, k, s) in enumerate(CONV_LAYERS):
            blocks.append(_ConvBlock(cin, c, k, s, group_norm=(i == 0)))
            cin = c
        self.conv_layers = nn.ModuleList(blocks)

    def forward(self, wav: torch.Tensor) -> torch.Tensor:        # (B, samples) -> (B, 512, frames)
        x = wav.unsqueeze(1)
        for blk in self.conv_layers:
            x = blk(x)
        return x


class SelfAttention(nn.Module):
    """fairseq MultiheadAttention, self-attention case: separate q / k / v projections with bias, scaling 1/sqrt(head_dim)"""

    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.heads = heads
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(dim, dim) for _ in range(4))

    def forward(self, x: torch.Tensor) -> torch.Tensor:           # (B, T, C)
        B, T, C = x.shape
        q, k, v = (p(x).view(B, T, self.heads, C // self.heads).transpose(1, 2) for p in (self.q_proj, self.k_proj, self.v_proj))
        y = F.scaled_dot_product_attention(q, k, v)                # scale = 1/sqrt(head_dim), as fairseq's `scaling`
        return self.out_proj(y.transpose(1, 2).reshape(B, T, C))


class EncoderLayer(nn.Module):
    """TransformerSentenceEncoderLayer with layer_norm_first = False (HuBERT base)"""

    def __init__(self, dim: int = 768, ffn: int = 3072, heads: int = 12):
        super().__init__()
        self.self_attn = SelfAttention(dim, heads)
        self.self_attn_layer_norm = nn.LayerNorm(dim)
        self.fc1, self.fc2 = nn.Linear(dim, ffn), nn.Linear(ffn, dim)
        self.final_layer_norm = nn.LayerNorm(dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.self_attn_layer_norm(x + self.self_attn(x))
        return self.final_layer_norm(x + self.fc2(F.gelu(self.fc1(x))))


class PosConv(nn.Module):
    """``encoder.pos_conv.0``: weight-normalised grouped conv; parameters kept in fairseq's (weight_g, weight_v) form"""

    def __init__(self, dim: int = 768, kernel: int = 128, groups: int = 16):
        super().__init__()
        self.kernel, self.groups = kernel, groups
        self.bias = nn.Parameter(torch.zeros(dim))
        self.weight_g = nn.Parameter(torch.ones(1, 1, kernel))                     # norm over dims (0, 1) per kernel tap: weight_norm(dim=2)
        self.weight_v = nn.Parameter(torch.randn(dim, dim // groups, kernel) * (4.0 / (kernel * dim)) ** 0.5)

    def weight(self) -> torch.Tensor:
        v = self.weight_v
        return v * (self.weight_g / v.norm(p=2, dim=(0, 1), keepdim=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:           # (B, C, T)
        y = F.conv1d(x, self.weight(), self.bias, padding=self.kernel // 2, groups=self.groups)
        return F.gelu(y[:, :, :-1] if self.kernel % 2 == 0 else y)               # SamePad: an even kernel yields one frame too many


class Encoder(nn.Module):
    def __init__(self, dim: int = 768, layers: int = 12):
        super().__init__()
        self.pos_conv = nn.ModuleList([PosConv(dim)])             # (fairseq: nn.Sequential(conv, SamePad, GELU) -> key "pos_conv.0.*")
        self.layers = nn.ModuleList([EncoderLayer(dim) for _ in range(layers)])
        self.layer_norm = nn.LayerNorm(dim)

    def forward(self, x: torch.Tensor, output_layer: int) -> torch.Tensor:        # (B, T, C)
        x = x + self.pos_conv[0](x.transpose(1, 2)).transpose(1, 2)
        x = self.layer_norm(x)
        for layer in self.layers[:output_layer]:
            x = layer(x)
        return x


class ContentVec(nn.Module):
    """``extract(wav16k (B, samples)) -> (B, 256, frames)``: the reference's ``utils.get_hubert_content`` (batched)"""

    def __init__(self, out_dim: int = 256):
        super().__init__()
        self.feature_extractor = ConvFeatureExtractor()
        self.layer_norm = nn.LayerNorm(512)
        self.post_extract_proj = nn.Linear(512, 768)
        self.encoder = Encoder()
        self.final_proj = nn.Linear(768, out_dim)

    @staticmethod
    def frames_for(samples: int) -> int:
        n = samples
        for _, k, s in CONV_LAYERS:
            n = (n - k) // s + 1
        return n

    def load_fairseq_state_dict(self, state: Dict[str, torch.Tensor]) -> None:
        """the ``model`` entry of ``checkpoint_best_legacy_500.pt``: everything this module owns must be there; the pre-training
        leftovers (``mask_emb``, ``label_embs_concat``) are not part of feature extraction"""
        own = {k: v for k, v in state.items() if k not in ("mask_emb", "label_embs_concat")}
        self.load_state_dict(own, strict=True)

    @torch.no_grad()
    def extract(self, wav16k: torch.Tensor, output_layer: int = 12, autocast=None) -> torch.Tensor:
        if wav16k.dim() == 1:
            wav16k = wav16k.unsqueeze(0)

        def run(w):
            f = self.feature_extractor(w).transpose(1, 2)                       # (B, frames, 512)
            x = self.post_extract_proj(self.layer_norm(f))
            return self.final_proj(self.encoder(x, output_layer))
        if autocast is not None and wav16k.is_cuda:
            with torch.autocast("cuda", dtype=autocast):
                y = run(wav16k.float())
        else:
            y = run(wav16k.float())
        return y.float().transpose(1, 2)

    def content(self, wav16k: torch.Tensor, target_frames: int, autocast=None) -> torch.Tensor:
        """``infer_tool.py:163-166``: the features stretched to the latent frame count (``repeat_expand_2d``) -> (B, 256, target_frames)"""
        from .audio import repeat_expand_2d
        return repeat_expand_2d(self.extract(wav16k, autocast=autocast), target_frames)

    forward = extract
