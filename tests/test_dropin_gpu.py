"""GPU tests of the torch-facing surfaces: the drop-in ``unet1d.UNet1DConditionModel`` nn.Module and
``ns2vc_amd.pipeline.Denoiser`` against the reference goldens (same procedural weights / inputs)."""
from __future__ import annotations

import os

import numpy as np
import pytest

from util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")


def _inputs(tag, B, T, Lp):
    import torch
    from ns2vc_amd.weights import hash_normal
    return (torch.from_numpy(hash_normal(f"{tag}.x", (B, 100, T))).cuda(), torch.from_numpy(hash_normal(f"{tag}.content", (B, 256, T))).cuda(),
            torch.from_numpy(hash_normal(f"{tag}.prompt", (B, Lp, 256))).cuda())


@pytest.fixture(scope="module")
def state():
    import torch
    from ns2vc_amd.weights import procedural_state_dict
    return {k: torch.from_numpy(v) for k, v in procedural_state_dict(seed=0).items()}


def test_dropin_module_forward_matches_reference_golden(state, diag):
    import torch
    from unet1d import UNet1DConditionModel
    gold = np.load(GOLD)
    m = UNet1DConditionModel(in_channels=356, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                             cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text",
                             resnet_time_scale_shift="scale_shift", engine_precision="fp32")
    m.load_state_dict(state, strict=True)          # the reference's load path (infer_tool.py:24-29)
    m = m.cuda().eval()
    x, content, prompt = _inputs("g3b", 2, 37, 21)
    mask = (torch.arange(21)[None, :] < torch.tensor([21, 13])[:, None]).cuda()
    with torch.no_grad():
        out = m(torch.cat([x, content], dim=1), torch.tensor([499.50003, 499.50003]).cuda(), prompt, encoder_attention_mask=mask)
        tup = m(torch.cat([x, content], dim=1), 499.50003, prompt, encoder_attention_mask=mask, return_dict=False)
    e = rel_l2(out.sample.cpu().numpy(), gold["g3b.y"])
    diag(f"drop-in nn.Module forward (fp32 engine) vs reference golden: {e:.3e}")
    assert e < 1e-3 and isinstance(tup, tuple) and rel_l2(tup[0].cpu().numpy(), gold["g3b.y"]) < 1e-3
    with pytest.raises(NotImplementedError):       # autograd / training is out of scope and must fail loudly
        m(torch.cat([x, content], dim=1), 3, prompt)
    # weights edited in place are picked up (the engine re-packs when a parameter version changes)
    with torch.no_grad():
        m.conv_out.bias.add_(1.0)
        out2 = m(torch.cat([x, content], dim=1), 499.50003, prompt, encoder_attention_mask=mask).sample
    assert abs(float((out2 - out.sample).mean()) - 1.0) < 1e-3


def test_pipeline_sampler_matches_reference_golden(state, diag):
    import torch
    from ns2vc_amd.pipeline import Denoiser
    gold = np.load(GOLD)
    d = Denoiser(state, precision="fp32")
    for tag, solver, steps, B in (("unipc6_b2", "unipc", 6, 2), ("dpm6_b3", "dpmsolver++", 6, 3)):
        xT, content, prompt = _inputs(f"g5.{tag}", B, 188, 469)
        mask = (torch.arange(469)[None, :] < torch.from_numpy(gold[f"g5.{tag}.lens"])[:, None]).cuda()
        y = d.sample(content, prompt, mask, noise=xT, solver=solver, steps=steps)
        e = rel_l2(y.cpu().numpy(), gold[f"g5.{tag}.y"])
        diag(f"pipeline.Denoiser.sample {tag}: {e:.3e}")
        assert e < 1e-3
    y1 = d.denoise(xT, torch.full((3,), 666.0).cuda(), content, prompt, mask)
    assert y1.shape == xT.shape and bool(torch.isfinite(y1).all())


def test_overlapped_pipeline_matches_sequential(diag):
    """pre(k+1) | denoise(k) | post(k-1) on three streams == the same stages run one after another (bit-identical),
    and the overlapped schedule is not slower."""
    import time
    import torch
    from ns2vc_amd.pipeline import Denoiser, OverlappedPipeline
    from ns2vc_amd.weights import procedural_state_dict
    dev = torch.device("cuda", 0)
    den = Denoiser(procedural_state_dict(seed=0), precision="bf16")
    B, T, Lp = 2, 188, 64
    Wpre = torch.randn(256, 256, device=dev) / 16.0
    Wpost = torch.randn(100, 100, device=dev) / 10.0

    def pre_fn(seed):                      # stand-in for ContentVec + Pre_model.infer: seeded tensors + some real work
        g = torch.Generator(device=dev).manual_seed(1000 + seed)
        c = torch.randn((B, 256, T), device=dev, generator=g)
        p = torch.randn((B, Lp, 256), device=dev, generator=g)
        for _ in range(20):
            p = torch.tanh(p @ Wpre)
        c = torch.tanh(torch.einsum("oc,bct->bot", Wpre, c))
        n = torch.randn((B, 100, T), device=dev, generator=g)
        m = torch.ones((B, Lp), dtype=torch.bool, device=dev)
        m[1, Lp // 2:] = False
        return {"content": c, "prompt": p, "prompt_mask": m, "noise": n}

    def post_fn(latent, seed):             # stand-in for vocos.decode
        y = latent
        for _ in range(20):
            y = torch.tanh(torch.einsum("oc,bct->bot", Wpost, y))
        return y

    items = list(range(5))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seq = []
    for k in items:
        cond = pre_fn(k)
        lat = den.sample(cond["content"], cond["prompt"], cond["prompt_mask"], cond["noise"], solver="unipc", steps=6, order=2)
        seq.append(post_fn(lat, k))
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    pipe = OverlappedPipeline(den, pre_fn, post_fn, solver="unipc", steps=6, order=2)
    pipe.run(items[:1])                    # warm the three streams / allocator pools
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.run(items)
    t_ovl = time.perf_counter() - t0
    assert len(out) == len(seq)
    for a, b in zip(out, seq):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)
    diag(f"overlapped pipeline: 5 batches sequential {t_seq * 1e3:.1f} ms, three-stream {t_ovl * 1e3:.1f} ms")
