"""GPU tests of the stages either side of the denoiser: the PyTorch-ROCm ``PreModel`` on the
device against the reference golden, the real front end inside the three-stream pipeline, and the batched converter that
groups equal-length segments (reference: one segment per call, inference/infer_tool.py:189-206)."""
import json
import os
import time

import numpy as np
import pytest

from util import procedural_params, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRE_CFG = {"phoneme_encoder": {"in_channels": 256, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2},
           "prompt_encoder": {"in_channels": 100, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2}}


@pytest.fixture(scope="module")
def weights():
    from ns2vc_amd.weights import procedural_state_dict
    return procedural_state_dict(seed=0)


@pytest.fixture(scope="module")
def pre_model():
    import torch
    from ns2vc_amd.frontend import PreModel
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "pre_model_state_keys.json")))
    m = PreModel(PRE_CFG).eval()
    m.load_state_dict(procedural_params(keys["keys"], "pre"), strict=True)
    return m.to(torch.device("cuda", 0))


def _segments(lengths, refer_lengths, tag="svc"):
    import torch
    from ns2vc_amd.service import Segment
    from ns2vc_amd.weights import hash_normal
    return [Segment(torch.from_numpy(hash_normal(f"{tag}.c{i}", (256, T))), torch.from_numpy(hash_normal(f"{tag}.r{i}", (100, L))), tag=i)
            for i, (T, L) in enumerate(zip(lengths, refer_lengths))]


def test_frontend_on_device_matches_reference_golden(pre_model, diag):
    """the conditioning front end on the MI355X (rocBLAS / SDPA kernels) against the reference's Pre_model.infer outputs"""
    import torch
    from ns2vc_amd.weights import hash_normal
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
    dev = torch.device("cuda", 0)
    B, T, Lp = 2, 65, 40
    lengths, rlens = torch.from_numpy(g["g10.lengths"]).to(dev), torch.from_numpy(g["g10.refer_lengths"]).to(dev)
    c = torch.from_numpy(hash_normal("g10.c", (B, 256, T))).to(dev) * (torch.arange(T, device=dev)[None, None, :] < lengths[:, None, None])
    refer = torch.from_numpy(hash_normal("g10.refer", (B, 100, Lp))).to(dev) * (torch.arange(Lp, device=dev)[None, None, :] < rlens[:, None, None])
    content, prompt, mask = pre_model.infer(c, refer, lengths, rlens)
    e = rel_l2(content.cpu().numpy(), g["g10.content"]), rel_l2(prompt.cpu().numpy(), g["g10.prompt"])
    diag(f"frontend on device vs reference Pre_model.infer: content {e[0]:.2e} prompt {e[1]:.2e}")
    assert max(e) < 1e-4                                  # fp32 library kernels: accumulation order differs from the CPU reference
    assert float(content[1, :, 50:].abs().max()) == 0.0 and float(prompt[1, 27:].abs().max()) == 0.0 and int(mask[1].sum()) == 27


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("fp16", 2e-3)])
def test_grouped_converter_equals_per_segment_runs(pre_model, precision, tol, diag):
    """batched conversion == one-segment-at-a-time conversion (the reference's batch-1 loop): segments are grouped by
    (latent length, prompt length) and never padded, so every segment keeps its batch-1 semantics.
    fp32 agrees to accumulation noise; the 16-bit mode to its rounding noise (a batched and a single run round
    differently: DESIGN.md 'batch independence')."""
    import torch
    from ns2vc_amd.pipeline import Denoiser
    from ns2vc_amd.service import GroupedConverter
    from ns2vc_amd.weights import procedural_state_dict
    den = Denoiser(procedural_state_dict(seed=0), precision=precision)
    lengths = [96, 130, 96, 64, 130, 96, 96]
    rlens = [40, 64, 40, 64, 64, 40, 50]
    segs = _segments(lengths, rlens)
    conv = GroupedConverter(pre_model, den, max_batch=3, solver="unipc", steps=6)
    groups = conv.plan(segs)
    assert groups == [[1, 4], [6], [0, 2, 5], [3]]                      # longest first, at most max_batch, input order inside
    out = conv.convert(segs)
    one = GroupedConverter(pre_model, den, max_batch=1, solver="unipc", steps=6).convert(segs)
    errs = []
    for i, (a, b) in enumerate(zip(out, one)):
        assert a.shape == (100, lengths[i]) and torch.isfinite(a).all()
        errs.append(rel_l2(a.cpu().numpy(), b.cpu().numpy()))
    diag(f"grouped converter ({precision}) vs per-segment runs: max rel {max(errs):.3e}")
    assert max(errs) < tol


def test_grouped_converter_segment_vs_oracle(pre_model, weights, diag):
    """one converted segment against the CPU oracle's sampler fed the same front-end outputs: the service adds grouping and
    stream plumbing, not arithmetic"""
    import torch
    from ns2vc_amd.pipeline import Denoiser
    from ns2vc_amd.service import GroupedConverter
    from ns2vc_amd.spec import UNetConfig
    from oracle import sampler_ref, unet_ref
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    den = Denoiser(weights, precision="fp32")
    segs = _segments([80, 80, 112], [48, 48, 20], tag="svc2")
    out = GroupedConverter(pre_model, den, max_batch=4, solver="unipc", steps=5, seed=7).convert(segs)
    i, T = 1, 80
    dev = torch.device("cuda", 0)
    content, prompt, mask = pre_model.infer(segs[i].content[None].to(dev), segs[i].refer[None].to(dev), torch.tensor([T], device=dev),
                                            torch.tensor([48], device=dev))
    P = {k: torch.from_numpy(v) for k, v in weights.items()}
    xT = torch.randn((100, T), generator=torch.Generator().manual_seed(7 + i))[None]
    tc, tp, tm = content.cpu(), prompt.cpu(), mask.cpu()
    ref = sampler_ref.unipc_bh2(lambda xx, tt: unet_ref.denoiser(P, UNetConfig(), xx, tc, tp, tm, tt), sampler_ref.linear_betas(1000), xT, 5)
    e = rel_l2(out[i].cpu().numpy(), ref[0].numpy())
    diag(f"grouped converter segment vs oracle sampler (fp32 engine): {e:.3e}")
    assert e < 1e-4


def test_pipeline_with_real_front_end_end_to_end_rtf(pre_model, diag):
    """BASELINE config-3 shape (32 x 10 s, 20-step UniPC) with the REAL conditioning front end as the pipeline's first stage:
    end-to-end (front end + denoiser) wall time per batch beside the denoiser alone, sequential and stream-overlapped."""
    import torch
    from ns2vc_amd.pipeline import Denoiser, OverlappedPipeline
    from ns2vc_amd.weights import procedural_state_dict
    dev = torch.device("cuda", 0)
    den = Denoiser(procedural_state_dict(seed=0))
    B, T, Lp, steps, n_batches = 32, 938, 469, 20, 4
    g = torch.Generator(device=dev).manual_seed(5)
    c = torch.randn((B, 256, T), device=dev, generator=g)
    refer = torch.randn((B, 100, Lp), device=dev, generator=g)
    lengths, rlens = torch.full((B,), T, device=dev), torch.full((B,), Lp, device=dev)
    noise = torch.randn((B, 100, T), device=dev, generator=g)

    def pre_fn(k):
        content, prompt, mask = pre_model.infer(c, refer, lengths, rlens)
        return {"content": content, "prompt": prompt, "prompt_mask": mask, "noise": noise}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    cond = pre_fn(0)
    den.sample(**{"content": cond["content"], "prompt": cond["prompt"], "prompt_mask": cond["prompt_mask"], "noise": noise}, solver="unipc", steps=steps)
    _, t_pre = timed(lambda: [pre_fn(k) for k in range(n_batches)])
    _, t_den = timed(lambda: [den.sample(cond["content"], cond["prompt"], cond["prompt_mask"], noise, solver="unipc", steps=steps) for _ in range(n_batches)])

    def sequential():
        outs = []
        for k in range(n_batches):
            cd = pre_fn(k)
            outs.append(den.sample(cd["content"], cd["prompt"], cd["prompt_mask"], cd["noise"], solver="unipc", steps=steps))
        return outs
    seq, t_seq = timed(sequential)
    pipe = OverlappedPipeline(den, pre_fn, lambda latent, k: latent, solver="unipc", steps=steps)
    pipe.run([0])
    ovl, t_ovl = timed(lambda: pipe.run(list(range(n_batches))))
    for a, b in zip(ovl, seq):
        assert torch.equal(a, b)
    audio_s = n_batches * B * T * 256 / 24000.0
    diag(f"end-to-end, {n_batches} batches of 32 x 10 s, 20-step UniPC: front end alone {t_pre / n_batches * 1e3:.1f} ms/batch, denoiser alone "
         f"{t_den / n_batches * 1e3:.1f} ms/batch, sequential {t_seq / n_batches * 1e3:.1f} ms/batch (RTF {t_seq / audio_s:.2e}), "
         f"overlapped {t_ovl / n_batches * 1e3:.1f} ms/batch (RTF {t_ovl / audio_s:.2e}); denoiser-only RTF {t_den / audio_s:.2e}")
    assert t_ovl < 1.30 * t_seq          # (a sanity bound only: wall-clock on a shared box; the numbers are in the diagnostics)
    # the same front end with 16-bit operands (torch.autocast): time, and what it does to the conditioning and the sampled latent
    def pre16(k):
        content, prompt, mask = pre_model.infer(c, refer, lengths, rlens, autocast=torch.float16)
        return {"content": content, "prompt": prompt, "prompt_mask": mask, "noise": noise}
    c16 = pre16(0)
    _, t_pre16 = timed(lambda: [pre16(k) for k in range(n_batches)])
    pipe16 = OverlappedPipeline(den, pre16, lambda latent, k: latent, solver="unipc", steps=steps)
    pipe16.run([0])
    ovl16, t_ovl16 = timed(lambda: pipe16.run(list(range(n_batches))))
    e_c, e_p = rel_l2(c16["content"].cpu().numpy(), cond["content"].cpu().numpy()), rel_l2(c16["prompt"].cpu().numpy(), cond["prompt"].cpu().numpy())
    e_y = rel_l2(ovl16[0].cpu().numpy(), seq[0].cpu().numpy())
    diag(f"  fp16-autocast front end: {t_pre16 / n_batches * 1e3:.1f} ms/batch alone, overlapped end-to-end {t_ovl16 / n_batches * 1e3:.1f} ms/batch "
         f"(RTF {t_ovl16 / audio_s:.2e}); vs the fp32 front end: content {e_c:.2e}, prompt {e_p:.2e}, sampled latent {e_y:.2e}")
    assert e_c < 5e-3 and e_p < 5e-3


def test_pipeline_end_to_end_with_the_vocoder_stage(pre_model, diag):
    """All three stages of the north_star's pipeline as REAL stages: Pre_model.infer (PyTorch-ROCm) | denoiser (HIP engine) |
    Vocos backbone + inverse STFT (PyTorch-ROCm, ns2vc_amd/vocoder.py: restated architecture, procedural weights, parity
    unpinned) on three streams, 32 x 10 s batches, 20-step UniPC.  The overlapped schedule returns the same waveforms as the
    stages run one after another; the end-to-end RTF now includes the vocoder, in fp32 and with 16-bit autocast for the two
    PyTorch stages."""
    import torch
    from ns2vc_amd.pipeline import Denoiser, OverlappedPipeline
    from ns2vc_amd.vocoder import VocosDecoder
    from ns2vc_amd.weights import procedural_state_dict
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    voc = VocosDecoder().eval().to(dev)
    den = Denoiser(procedural_state_dict(seed=0))
    B, T, Lp, steps, n_batches = 32, 938, 469, 20, 3
    g = torch.Generator(device=dev).manual_seed(5)
    c = torch.randn((B, 256, T), device=dev, generator=g)
    refer = torch.randn((B, 100, Lp), device=dev, generator=g)
    lengths, rlens = torch.full((B,), T, device=dev), torch.full((B,), Lp, device=dev)
    noise = torch.randn((B, 100, T), device=dev, generator=g)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    res = {}
    for name, ac in (("fp32", None), ("fp16 autocast", torch.float16)):
        def pre_fn(k):
            content, prompt, mask = pre_model.infer(c, refer, lengths, rlens, autocast=ac)
            return {"content": content, "prompt": prompt, "prompt_mask": mask, "noise": noise}

        def post_fn(latent, k):
            return voc.decode(latent, autocast=ac)

        def sequential():
            outs = []
            for k in range(n_batches):
                cd = pre_fn(k)
                outs.append(post_fn(den.sample(cd["content"], cd["prompt"], cd["prompt_mask"], cd["noise"], solver="unipc", steps=steps), k))
            return outs
        sequential()                                       # warm-up (plans, graph, library handles)
        cd0 = pre_fn(0)
        lat = den.sample(cd0["content"], cd0["prompt"], cd0["prompt_mask"], cd0["noise"], solver="unipc", steps=steps)
        _, t_voc = timed(lambda: [post_fn(lat, k) for k in range(n_batches)])
        seq, t_seq = timed(sequential)
        pipe = OverlappedPipeline(den, pre_fn, post_fn, solver="unipc", steps=steps)
        pipe.run([0])
        ovl, t_ovl = timed(lambda: pipe.run(list(range(n_batches))))
        for a, b in zip(ovl, seq):      # (the latents are bit-identical; the library kernels of the vocoder -- MIOpen / rocBLAS / the
            assert a.shape == (B, (T - 1) * 256) and bool(torch.isfinite(a).all())     # inverse STFT's overlap-add -- are not run-to-run deterministic)
            assert float((a - b).norm() / b.norm()) < (1e-4 if ac is None else 1e-2)
        audio_s = n_batches * B * T * 256 / 24000.0
        res[name] = (t_voc / n_batches * 1e3, t_seq / n_batches * 1e3, t_ovl / n_batches * 1e3, t_ovl / audio_s)
    diag("end-to-end WITH the vocoder stage (32 x 10 s, 20-step UniPC; Vocos restatement, procedural weights, parity unpinned): " +
         "; ".join(f"{k}: vocoder alone {v[0]:.1f} ms/batch, sequential {v[1]:.1f}, three streams {v[2]:.1f} ms/batch (RTF {v[3]:.2e})" for k, v in res.items()))
    assert res["fp32"][2] < 1.3 * res["fp32"][1]


@pytest.mark.gpu
def test_content_encoder_stage_from_waveform_to_sampled_latent(pre_model, diag):
    """The stage in FRONT of the conditioning front end: 16 kHz waveform -> ContentVec / HuBERT-base features (ns2vc_amd/contentvec.py:
    restated architecture, random weights, parity unpinned) -> repeat_expand_2d to the latent frame count (infer_tool.py:163-166)
    -> Pre_model.infer -> denoiser.  Checks the frame arithmetic on the device, that batched == per-utterance extraction, that the
    16-bit autocast stays close, that the result drives the rest of the path to a finite latent, and reports what the stage costs."""
    import torch
    from ns2vc_amd.contentvec import ContentVec
    from ns2vc_amd.pipeline import Denoiser
    from ns2vc_amd.weights import procedural_state_dict
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    enc = ContentVec().eval().to(dev)
    B, T, Lp, steps = 8, 938, 469, 4
    g = torch.Generator(device=dev).manual_seed(11)
    wav = 0.1 * torch.randn((B, 160000), device=dev, generator=g)                      # 10 s at 16 kHz
    feats = enc.extract(wav)
    assert feats.shape == (B, 256, ContentVec.frames_for(160000)) == (B, 256, 499) and torch.isfinite(feats).all()
    one = enc.extract(wav[3])
    e_batch = float((one - feats[3:4]).norm() / feats[3:4].norm())
    f16 = enc.extract(wav, autocast=torch.float16)
    e_f16 = float((f16 - feats).norm() / feats.norm())
    c = enc.content(wav, T)
    assert c.shape == (B, 256, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        enc.content(wav, T, autocast=torch.float16)
    torch.cuda.synchronize()
    ms16 = (time.perf_counter() - t0) / 3 * 1e3
    t0 = time.perf_counter()
    for _ in range(3):
        enc.content(wav, T)
    torch.cuda.synchronize()
    ms32 = (time.perf_counter() - t0) / 3 * 1e3
    refer = torch.randn((B, 100, Lp), device=dev, generator=g)
    lengths, rlens = torch.full((B,), T, device=dev), torch.full((B,), Lp, device=dev)
    content, prompt, mask = pre_model.infer(c, refer, lengths, rlens)
    den = Denoiser(procedural_state_dict(seed=0))
    lat = den.sample(content, prompt, mask, torch.randn((B, 100, T), device=dev, generator=g), solver="unipc", steps=steps)
    diag(f"content encoder stage: {B} x 10 s -> {tuple(feats.shape)}; batched vs single {e_batch:.1e}; fp16 autocast {e_f16:.1e}; "
         f"{ms32:.1f} ms fp32 / {ms16:.1f} ms fp16 autocast per batch of {B}; latent finite {bool(torch.isfinite(lat).all())}")
    assert e_batch < 1e-4 and e_f16 < 2e-2 and torch.isfinite(lat).all()
