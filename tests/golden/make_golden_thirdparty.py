#!/usr/bin/env python3
"""Pinning hook for the two third-party stages either side of the denoiser, for a machine that HAS what this
image lacks: `vocos` + its `charactr/vocos-mel-24khz` weights, `torchaudio`, `fairseq` + `hubert/checkpoint_best_legacy_500.pt`.

    python tests/golden/make_golden_thirdparty.py [--hubert hubert/checkpoint_best_legacy_500.pt] [--out tests/golden/golden_thirdparty.npz]

Each block runs the REAL package exactly as the reference calls it and stores inputs + outputs (never source or weights):
  mel.*        torchaudio.transforms.MelSpectrogram(24000, 1024, 256, 100, center=True, power=1) + log(clip(., 1e-7))    (infer_tool.py:170-182)
  vocos.*      Vocos.from_pretrained("charactr/vocos-mel-24khz").decode(mel)                                                 (model.py:689-691, 762)
  hubert.*     fairseq extract_features(output_layer=12) + final_proj, transposed                                            (utils.py:209-236)
Blocks whose packages are missing are skipped and reported; tests/test_cpu.py::test_thirdparty_goldens_* consume whatever the file
holds and are skipped while it does not exist.  The weights are needed again on the consuming side for vocos.* / hubert.*
(NS2VC_VOCOS_STATE = a torch-loadable state dict of the vocoder, NS2VC_HUBERT_CKPT = the fairseq checkpoint); mel.* needs nothing.
Until this has been run somewhere, f2 / f4 stay "parity unpinned" and every end-to-end RTF says so."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--hubert", default="hubert/checkpoint_best_legacy_500.pt")
    ap.add_argument("--out", default=os.path.join(HERE, "golden_thirdparty.npz"))
    a = ap.parse_args()
    out, report = {}, {}
    g = torch.Generator().manual_seed(20260927)
    wav24 = (0.3 * torch.randn(1, 24000, generator=g)).clamp(-1, 1)                      # 1 s of noise-like audio: every mel band is exercised
    wav16 = (0.3 * torch.randn(1, 16000 * 2, generator=g)).clamp(-1, 1)
    try:
        import torchaudio
        mel = torch.log(torch.clip(torchaudio.transforms.MelSpectrogram(sample_rate=24000, n_fft=1024, hop_length=256, n_mels=100, center=True, power=1)(wav24), min=1e-7))
        out["mel.wav24k"], out["mel.log_mel"] = wav24.numpy(), mel.numpy()
        report["mel"] = f"torchaudio {torchaudio.__version__}"
    except Exception as ex:
        report["mel"] = f"skipped: {ex!r}"
    try:
        from vocos import Vocos
        voc = Vocos.from_pretrained("charactr/vocos-mel-24khz").eval()
        mel_in = out["mel.log_mel"] if "mel.log_mel" in out else torch.randn(1, 100, 94, generator=g).numpy()
        with torch.no_grad():
            audio = voc.decode(torch.from_numpy(np.asarray(mel_in)))
        out["vocos.mel"], out["vocos.audio"] = np.asarray(mel_in), audio.numpy()
        report["vocos"] = f"{sum(p.numel() for p in voc.parameters())} parameters"
    except Exception as ex:
        report["vocos"] = f"skipped: {ex!r}"
    try:
        from fairseq import checkpoint_utils
        models, _, _ = checkpoint_utils.load_model_ensemble_and_task([a.hubert], suffix="")
        hm = models[0].eval()
        with torch.no_grad():
            logits = hm.extract_features(source=wav16, padding_mask=torch.zeros_like(wav16, dtype=torch.bool), output_layer=12)
            feats = hm.final_proj(logits[0]).transpose(1, 2)
        out["hubert.wav16k"], out["hubert.content"] = wav16.numpy(), feats.numpy()
        report["hubert"] = f"{a.hubert}: content {tuple(feats.shape)}"
    except Exception as ex:
        report["hubert"] = f"skipped: {ex!r}"
    for k, v in report.items():
        print(f"{k}: {v}")
    if not out:
        print("nothing to write: none of torchaudio / vocos / fairseq is importable here")
        return 1
    np.savez_compressed(a.out, **out)
    print(f"wrote {a.out}: {sorted(out)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
