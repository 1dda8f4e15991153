"""The two small signal-side helpers that sit between the audio files and the conditioning front end -- SURVEY 8(f) ranks 2 / 4.

* ``repeat_expand_2d`` (reference ``utils.py:482-496``): stretches the 50 Hz ContentVec features to the latent frame rate by
  repeating columns.  The reference walks the target frames in a Python loop with float32 comparisons; here the same walk
  produces an INDEX MAP once (numpy, float32 arithmetic as torch does it) and the expansion is one gather on the tensor's
  device.  Pinned bit-exact against the reference's own function (tests/golden ``g11.*``).
* ``log_mel`` (reference ``inference/infer_tool.py:170-182`` / ``preprocess.py:50-60``): the prompt's log-mel spectrogram,
  ``torchaudio.transforms.MelSpectrogram(24000, n_fft=1024, hop_length=256, n_mels=100, center=True, power=1)`` followed by
  ``log(clip(., 1e-7))``.  torchaudio is a pip dependency of the reference that is absent from this image, so this is a
  restatement of its published algorithm (torchaudio 2.x ``functional.spectrogram`` + ``melscale_fbanks``: periodic Hann
  window, reflect padding, magnitude spectrum, HTK mel scale, no filter normalisation, f_min 0, f_max sr/2) on
  ``torch.stft`` -- **parity unpinned** against torchaudio itself; tests check it against an independent numpy DFT.

The vocoder (``vocos.decode``, ``model.py:689-691``) and the ContentVec extractor are third-party models outside the
reference tree; their published architectures are restated in ``vocoder.py`` / ``contentvec.py`` (parity unpinned) and plug in
as ``decode_fn`` / as the producer of ``Segment.content`` (ns2vc_amd/service.py).
"""
from __future__ import annotations

from functools import lru_cache

import numpy as np
import torch


@lru_cache(maxsize=256)
def repeat_expand_index(src_len: int, target_len: int) -> np.ndarray:
    """source column of every target frame, exactly as the reference's loop picks it (utils.py:486-494)"""
    temp = (np.arange(src_len + 1, dtype=np.int64) * target_len).astype(np.float32) / np.float32(src_len)   # int64 * int / int -> float32
    idx = np.empty(target_len, dtype=np.int64)
    pos = 0
    for i in range(target_len):
        if not (np.float32(i) < temp[pos + 1]):
            pos += 1
        idx[i] = pos
    return idx


def repeat_expand_2d(content: torch.Tensor, target_len: int) -> torch.Tensor:
    """content (h, t) -> (h, target_len) float32 (reference ``utils.repeat_expand_2d``); also accepts (B, h, t)"""
    idx = torch.from_numpy(repeat_expand_index(int(content.shape[-1]), int(target_len))).to(content.device)
    return content.to(torch.float32).index_select(-1, idx)


@lru_cache(maxsize=8)
def mel_filterbank(n_freqs: int = 513, n_mels: int = 100, sample_rate: int = 24000, f_min: float = 0.0, f_max: float = None) -> torch.Tensor:
    """(n_freqs, n_mels) triangular HTK filters without normalisation (torchaudio.functional.melscale_fbanks defaults)"""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


def log_mel(wav: torch.Tensor, sample_rate: int = 24000, n_fft: int = 1024, hop_length: int = 256, n_mels: int = 100,
            floor: float = 1e-7) -> torch.Tensor:
    """wav (..., samples) at `sample_rate` -> log-mel (..., n_mels, 1 + samples // hop_length), what ``Pre_model.infer`` takes as
    ``refer_padded``"""
    window = torch.hann_window(n_fft, periodic=True, device=wav.device, dtype=wav.dtype)
    lead = wav.shape[:-1]
    spec = torch.stft(wav.reshape(-1, wav.shape[-1]), n_fft, hop_length=hop_length, win_length=n_fft, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()            # power = 1
    fb = mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate).to(device=wav.device, dtype=wav.dtype)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    return torch.log(torch.clamp(mel, min=floor)).reshape(*lead, n_mels, -1)
