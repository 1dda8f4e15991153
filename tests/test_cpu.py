"""CPU suite (no GPU): the oracle against the reference's golden vectors, the host
logic (parameter spec, procedural weights, solver tables, sharding), and that the
C-ABI library loads and exports every symbol the header declares."""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "golden_v1.npz"))


@pytest.fixture(scope="module")
def weights():
    from ns2vc_amd.weights import procedural_state_dict
    return {k: torch.from_numpy(v) for k, v in procedural_state_dict(seed=0).items()}


def inputs(tag, B, T, Lp):
    from ns2vc_amd.weights import hash_normal
    return (torch.from_numpy(hash_normal(f"{tag}.x", (B, 100, T))), torch.from_numpy(hash_normal(f"{tag}.content", (B, 256, T))),
            torch.from_numpy(hash_normal(f"{tag}.prompt", (B, Lp, 256))))


# ---- spec / weights -------------------------------------------------------------------
def test_param_spec_matches_reference_state_dict():
    from ns2vc_amd.spec import UNetConfig, param_spec
    ref = json.load(open(os.path.join(GOLD, "unet_state_keys.json")))
    spec = param_spec(UNetConfig())
    assert [[k, list(s)] for k, s in spec.items()] == ref["keys"]
    assert ref["n_tensors"] == 701 and ref["n_params"] == 66076900      # demo.ipynb:448


def test_dropin_module_state_dict_and_loud_failures():
    from unet1d import UNet1DConditionModel
    from unet1d.embeddings import TextTimeEmbedding          # model.py:6 import must keep working
    m = UNet1DConditionModel(in_channels=356, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                             cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text",
                             resnet_time_scale_shift="scale_shift")
    ref = json.load(open(os.path.join(GOLD, "unet_state_keys.json")))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["keys"]
    with torch.no_grad(), pytest.warns(UserWarning, match="slow plumbing path"):      # CPU tensors: the module's own PyTorch path, loudly
        y = m(torch.zeros(1, 356, 8), 3, torch.zeros(1, 4, 256)).sample
    assert y.shape == (1, 100, 8) and m.cpu_calls == 1 and m.engine_calls == 0
    with pytest.raises(ValueError):
        UNet1DConditionModel(in_channels=356, out_channels=100, block_out_channels=(128, 256), norm_num_groups=8,
                             cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text", resnet_time_scale_shift="default")
    e = TextTimeEmbedding(100, 100, 1)
    assert e(torch.randn(2, 7, 100)).shape == (2, 100)


def test_procedural_weights_are_pinned():
    from ns2vc_amd.weights import hash_normal, hash_uniform, procedural_tensor
    u = hash_uniform("conv_in.weight", 5, seed=0)
    assert u.dtype == np.float32 and np.all(np.abs(u) <= 1)
    assert np.array_equal(u, hash_uniform("conv_in.weight", 5, seed=0))
    assert not np.array_equal(u, hash_uniform("conv_in.weight", 5, seed=1))
    w = procedural_tensor("down_blocks.0.resnets.0.norm1.weight", (128,))
    assert abs(float(w.mean()) - 1.0) < 0.05
    n = hash_normal("x", (4, 1000))
    assert abs(float(n.std()) - 1.0) < 0.05 and abs(float(n.mean())) < 0.05


def test_checkpoint_layout_roundtrip(tmp_path):
    from ns2vc_amd.spec import UNetConfig, param_spec
    from ns2vc_amd.weights import load_checkpoint, save_checkpoint, unet_state_from_checkpoint
    spec = param_spec(UNetConfig())
    sd = {k: torch.zeros(s) for k, s in spec.items()}
    sd["conv_in.bias"] += 3
    p = str(tmp_path / "model-1.pt")
    save_checkpoint(p, sd, step=1000, extra={"pre_model.dummy": torch.zeros(1)})
    raw = torch.load(p, map_location="cpu", weights_only=False)
    assert set(raw.keys()) == {"step", "model"} and all(k.startswith(("diff_model.unet.", "pre_model.")) for k in raw["model"])
    back = load_checkpoint(p)
    assert list(back.keys()) == list(spec.keys()) and float(back["conv_in.bias"][0]) == 3
    with pytest.raises(KeyError):
        unet_state_from_checkpoint({"model": {"diff_model.unet.conv_in.weight": sd["conv_in.weight"]}})


def test_flop_model_matches_published():
    from ns2vc_amd.spec import PUBLISHED_GFLOP, algorithmic_gflop_per_sample_step, frames_for_seconds
    assert [frames_for_seconds(s) for s in (2, 10, 30)] == [188, 938, 2813]
    for (T, Lp), g in PUBLISHED_GFLOP.items():
        assert abs(algorithmic_gflop_per_sample_step(T, Lp) / g - 1) < 2e-3, (T, Lp)


# ---- oracle vs the reference goldens --------------------------------------------------------
def test_oracle_forward_golden_odd_T_ragged(gold, weights):
    from ns2vc_amd.spec import UNetConfig
    from oracle import unet_ref
    x, content, prompt = inputs("g3b", 2, 37, 21)
    mask = torch.arange(21)[None, :] < torch.tensor([21, 13])[:, None]
    y = unet_ref.unet_forward(weights, UNetConfig(), torch.cat([x, content], 1), torch.tensor([499.50003, 499.50003]), prompt, mask)
    assert y.shape == (2, 100, 37)
    assert rel_l2(y, gold["g3b.y"]) < 1e-5


@pytest.mark.parametrize("tag,tval", [("t999", torch.tensor([999.0])), ("t166_5", torch.tensor([166.5])), ("t3_int", torch.tensor([3]))])
def test_oracle_forward_golden_2s_with_block_checksums(gold, weights, tag, tval):
    from ns2vc_amd.spec import UNetConfig
    from oracle import unet_ref
    x, content, prompt = inputs("g2", 1, 188, 469)
    taps = {}
    y = unet_ref.unet_forward(weights, UNetConfig(), torch.cat([x, content], 1), tval, prompt, torch.ones(1, 469, dtype=torch.bool), taps=taps)
    assert rel_l2(y, gold[f"g2.{tag}.y"]) < 1e-5
    names = [str(n) for n in gold[f"g2.{tag}.tap_names"]]
    chk = gold[f"g2.{tag}.taps"]
    for i, n in enumerate(names):          # per-block mean / std / absmax recorded from the reference's own modules
        v = taps[n].double()
        got = np.array([float(v.mean()), float(v.std()), float(v.abs().max())])
        assert np.allclose(got, chk[i, :3], rtol=1e-4, atol=1e-5), n


def test_oracle_adapter_and_mask_goldens(gold, weights):
    from ns2vc_amd.spec import UNetConfig
    from oracle import unet_ref
    cfg = UNetConfig()
    x, content, prompt = inputs("g3", 2, 188, 469)
    mask = torch.arange(469)[None, :] < torch.from_numpy(gold["g3.lens"])[:, None]
    t = torch.tensor([832.50006, 832.50006])
    assert rel_l2(unet_ref.denoiser(weights, cfg, x, content, prompt, mask, t), gold["g3.ragged.y"]) < 1e-5
    assert rel_l2(unet_ref.denoiser(weights, cfg, x, content, prompt, None, t), gold["g3.nomask.y"]) < 1e-5
    if "g5b.y" in gold:
        x, content, prompt = inputs("g5b", 2, 188, 469)
        mask = torch.arange(469)[None, :] < torch.tensor([469, 300])[:, None]
        assert rel_l2(unet_ref.denoiser(weights, cfg, x, content, prompt, mask, torch.tensor([666.0, 666.0])), gold["g5b.y"]) < 1e-5


@pytest.mark.parametrize("tag,kind,steps,B,order", [("dpm6_b3", "dpm", 6, 3, 2), ("dpm1_b1_plumbing", "dpm", 1, 1, 1), ("unipc6_b2", "unipc", 6, 2, 2)])
def test_oracle_sampler_and_tables_vs_reference_golden(gold, weights, tag, kind, steps, B, order):
    """The oracle samplers AND the product's host-precomputed coefficient tables both reproduce the reference loop."""
    from ns2vc_amd import schedule as S
    from ns2vc_amd.spec import UNetConfig
    from oracle import sampler_ref, unet_ref
    cfg = UNetConfig()
    xT, content, prompt = inputs(f"g5.{tag}", B, 188, 469)
    mask = torch.arange(469)[None, :] < torch.from_numpy(gold[f"g5.{tag}.lens"])[:, None]
    betas = sampler_ref.linear_betas()

    def x0(xx, tt):
        return unet_ref.denoiser(weights, cfg, xx, content, prompt, mask, tt)

    if kind == "dpm":
        y = sampler_ref.dpm_solver_pp_2m(x0, betas, xT, steps, order)
        table = S.build_table("dpmsolver++", steps, betas.numpy(), order)
    else:
        y = sampler_ref.unipc_bh2(x0, betas, xT, steps)
        table = S.build_table("unipc", steps, betas.numpy(), order)
    assert rel_l2(y, gold[f"g5.{tag}.y"]) < 1e-5
    y_tab = S.run_table_numpy(table, lambda a, t: x0(torch.from_numpy(a), torch.from_numpy(t)).numpy(), xT.numpy())
    assert rel_l2(y_tab, gold[f"g5.{tag}.y"]) < 1e-4


# ---- host solver tables -------------------------------------------------------------------
@pytest.mark.parametrize("steps", [1, 6, 20, 30, 40, 50])
def test_schedule_scalars_match_reference(gold, steps):
    from ns2vc_amd import schedule as S
    from oracle import sampler_ref
    sched = S.VPSchedule(S.linear_betas())
    ts = sched.timesteps(steps)
    assert np.allclose(ts, gold[f"g4.s{steps}.t"], rtol=0, atol=1e-7)
    assert np.allclose([sched.lam(t) for t in ts], gold[f"g4.s{steps}.lambda"], rtol=2e-5, atol=2e-6)
    assert np.allclose([sched.alpha(t) for t in ts], gold[f"g4.s{steps}.alpha"], rtol=2e-5, atol=1e-7)
    assert np.allclose([sched.sigma(t) for t in ts], gold[f"g4.s{steps}.sigma"], rtol=2e-5, atol=1e-7)
    assert np.allclose([sched.model_time(t) for t in ts], gold[f"g4.s{steps}.t_model"], rtol=0, atol=1e-3)
    osched = sampler_ref.VPSchedule(sampler_ref.linear_betas())
    assert np.allclose(osched.lam(torch.from_numpy(gold[f"g4.s{steps}.t"])).numpy(), gold[f"g4.s{steps}.lambda"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("solver,steps", [("dpmsolver++", 2), ("dpmsolver++", 9), ("dpmsolver++", 10), ("dpmsolver++", 50),
                                          ("unipc", 2), ("unipc", 3), ("unipc", 20), ("unipc", 30)])
def test_tables_reproduce_oracle_samplers_on_a_toy_model(solver, steps):
    """Unified recurrence == the literal solver loops, for every step count regime
    (lower_order_final for <10 DPM steps, UniPC's order-1 first/last steps), batch 3."""
    from ns2vc_amd import schedule as S
    from oracle import sampler_ref
    rng = np.random.default_rng(steps)
    xT = rng.standard_normal((3, 5, 11)).astype(np.float32)
    Wm = rng.standard_normal((5, 5)).astype(np.float32) * 0.3

    def x0_t(x, t):          # a smooth nonlinear stand-in for the denoiser
        return torch.tanh(torch.einsum("oc,bct->bot", torch.from_numpy(Wm), x)) * (1.0 - t[:, None, None] / 2000.0)

    betas = sampler_ref.linear_betas()
    if solver == "unipc":
        ref = sampler_ref.unipc_bh2(x0_t, betas, torch.from_numpy(xT), steps)
    else:
        ref = sampler_ref.dpm_solver_pp_2m(x0_t, betas, torch.from_numpy(xT), steps, 2)
    table = S.build_table(solver, steps, betas.numpy(), 2)
    assert table.coef.shape == (steps, S.NCOEF) and table.coef.dtype == np.float32
    got = S.run_table_numpy(table, lambda a, t: x0_t(torch.from_numpy(a), torch.from_numpy(t)).numpy(), xT)
    assert rel_l2(got, ref.numpy()) < 5e-5


def test_table_argument_errors():
    from ns2vc_amd import schedule as S
    with pytest.raises(ValueError):
        S.build_table("ddim", 10)
    with pytest.raises(ValueError):
        S.build_table("unipc", 1, order=2)        # reference asserts steps >= order (dpm_solver.py:1172, uni_pc.py:607)
    assert S.build_table("dpmsolver++", 1, order=1).coef.shape == (1, S.NCOEF)


# ---- round-2 goldens (tests/golden/make_golden_v2.py): oracle and the PyTorch conditioning module vs the reference ----
def test_oracle_reproduces_golden_v2_forwards():
    """The oracle at the BASELINE lengths (10 s full output; 30 s windows + sums) against the reference's outputs."""
    import torch
    from ns2vc_amd.spec import UNetConfig
    from ns2vc_amd.weights import hash_normal, procedural_state_dict
    from oracle import unet_ref
    from util import g7_summary
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
    cfg = UNetConfig()
    P = {k: torch.from_numpy(v) for k, v in procedural_state_dict(cfg, 0).items()}
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for tag, T in (("g6", 938), ("g7", 2813)):
        x = torch.from_numpy(hash_normal(f"{tag}.x", (1, cfg.latent_channels, T)))
        content = torch.from_numpy(hash_normal(f"{tag}.content", (1, cfg.content_channels, T)))
        prompt = torch.from_numpy(hash_normal(f"{tag}.prompt", (1, 469, cfg.cross_attention_dim)))
        mask = torch.arange(469)[None, :] < torch.from_numpy(g[f"{tag}.prompt_len"])[:, None]
        y = unet_ref.denoiser(P, cfg, x, content, prompt, mask, torch.from_numpy(g[f"{tag}.t"])).numpy()
        if tag == "g6":
            assert rel_l2(y, g["g6.y"]) < 2e-6
        else:
            got = g7_summary(y)
            for k in ("head", "mid", "tail", "chan_sum", "chan_sq", "frame_sum", "frame_sq"):
                assert rel_l2(got[k], g[f"g7.{k}"]) < 2e-6, k


def test_text_time_embedding_matches_reference_golden():
    """unet1d.embeddings.TextTimeEmbedding (imported by the reference's model.py:6 for Pre_model.ref_enc, model.py:340:
    TextTimeEmbedding(100, 100, 1)) against outputs of the reference's own class on the same procedural parameters."""
    import torch
    from unet1d.embeddings import TextTimeEmbedding
    from ns2vc_amd.weights import hash_normal
    from util import tte_state
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
    for tag in ("g8.ref_enc", "g8.h64"):
        dim, odim, heads, L = (int(v) for v in g[f"{tag}.shape"])
        m = TextTimeEmbedding(dim, odim, heads).eval()
        m.load_state_dict(tte_state(tag, dim, odim), strict=True)       # reference parameter names load strict
        with torch.no_grad():
            y = m(torch.from_numpy(hash_normal(tag + ".x", (2, L, dim)))).numpy()
        assert y.shape == g[f"{tag}.y"].shape
        assert rel_l2(y, g[f"{tag}.y"]) < 1e-5, tag


def test_frontend_pre_model_matches_reference_golden():
    """ns2vc_amd.frontend.PreModel (the PyTorch-ROCm conditioning stage) against outputs of the
    reference's own Pre_model.infer (model.py:359-376) on a ragged batch: same state-dict names (strict load of the
    reference's key list), content / prompt within fp32 noise, padded frames exactly zero."""
    import json
    import torch
    from ns2vc_amd.frontend import PreModel
    from ns2vc_amd.weights import hash_normal
    from util import procedural_params
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "pre_model_state_keys.json")))
    cfg = {"phoneme_encoder": {"in_channels": 256, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2},
           "prompt_encoder": {"in_channels": 100, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2}}
    m = PreModel(cfg).eval()
    assert sorted(k for k, _ in keys["keys"]) == sorted(m.state_dict().keys())     # exactly the reference's 251 names
    assert keys["n_params"] == sum(p.numel() for p in m.parameters()) == 34923404     # demo.ipynb:447 "pre params"
    m.load_state_dict(procedural_params(keys["keys"], "pre"), strict=True)
    B, T, Lp = 2, 65, 40
    lengths, rlens = torch.from_numpy(g["g10.lengths"]), torch.from_numpy(g["g10.refer_lengths"])
    c = torch.from_numpy(hash_normal("g10.c", (B, 256, T))) * (torch.arange(T)[None, None, :] < lengths[:, None, None])
    refer = torch.from_numpy(hash_normal("g10.refer", (B, 100, Lp))) * (torch.arange(Lp)[None, None, :] < rlens[:, None, None])
    content, prompt, mask = m.infer(c, refer, lengths, rlens)
    assert content.shape == (B, 256, T) and prompt.shape == (B, Lp, 256) and mask.shape == (B, Lp)
    assert rel_l2(content.numpy(), g["g10.content"]) < 1e-5 and rel_l2(prompt.numpy(), g["g10.prompt"]) < 1e-5
    assert float(content[1, :, 50:].abs().max()) == 0.0 and float(prompt[1, 27:].abs().max()) == 0.0
    assert mask[1].sum() == 27


def test_dropin_training_path_matches_reference_and_backpropagates():
    """train.py drop-in (SURVEY 8(b) 'Threading / streams'): under autograd the drop-in module evaluates with PyTorch ops
    (unet1d/torch_path.py).  Forward == the reference golden (g3b: B=2, odd T, ragged mask), gradients reach every one of
    the 701 parameters, and the no-grad inference path still refuses CPU tensors (no CPU fallback)."""
    import torch
    from unet1d import UNet1DConditionModel
    from ns2vc_amd.weights import hash_normal, procedural_state_dict
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
    m = UNet1DConditionModel(in_channels=356, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                             cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text", resnet_time_scale_shift="scale_shift")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in procedural_state_dict(seed=0).items()}, strict=True)
    m.train()
    B, T, Lp = 2, 37, 21
    x = torch.from_numpy(hash_normal("g3b.x", (B, 100, T)))
    content = torch.from_numpy(hash_normal("g3b.content", (B, 256, T)))
    prompt = torch.from_numpy(hash_normal("g3b.prompt", (B, Lp, 256)))
    mask = torch.arange(Lp)[None, :] < torch.tensor([21, 13])[:, None]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out = m(torch.cat([x, content], dim=1), torch.tensor([499.50003, 499.50003]), prompt, encoder_attention_mask=mask).sample
    assert m.autograd_calls == 1 and m.engine_calls == 0
    assert rel_l2(out.detach().numpy(), gold["g3b.y"]) < 2e-5
    out.square().mean().backward()
    missing = [n for n, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing[:5]
    assert sum(float(p.grad.abs().sum()) > 0 for p in m.parameters()) >= 690      # (a few biases can see exactly zero gradient)
    m.eval()
    with pytest.warns(UserWarning, match="autograd is recording while the module is in eval"):      # an eval call that forgot no_grad()
        m(torch.cat([x, content], dim=1), 3, prompt, encoder_attention_mask=mask)
    assert m.autograd_calls == 2


def test_config1_cpu_plumbing_through_the_boundary():
    """BASELINE configs[0]: "2 s random Vocos latent + random ContentVec cond, 1 DPM-Solver step, batch 1 on PyTorch CPU
    (plumbing, no GPU)" -- the reference's `infer.py --device cpu` (inference/infer_tool.py:119-135) calls the UNet under
    torch.no_grad() with CPU tensors.  The drop-in module serves that call with its own PyTorch forward
    (unet1d/torch_path.py: product code, not the oracle) and says so once; one first-order DPM-Solver++ step driven the
    way the reference drives it (cat([x, content]) per evaluation, model.py:409; x_start wrapper, dpm_solver.py:271-292)
    reproduces the reference's own sampler output (golden g5.dpm1_b1_plumbing)."""
    import warnings
    from unet1d import UNet1DConditionModel
    from ns2vc_amd import schedule as S
    from ns2vc_amd.weights import procedural_state_dict
    gold = np.load(os.path.join(GOLD, "golden_v1.npz"))
    m = UNet1DConditionModel(in_channels=356, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                             cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text", resnet_time_scale_shift="scale_shift")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in procedural_state_dict(seed=0).items()}, strict=True)
    m.eval()
    xT, content, prompt = inputs("g5.dpm1_b1_plumbing", 1, 188, 469)
    mask = torch.arange(469)[None, :] < torch.from_numpy(gold["g5.dpm1_b1_plumbing.lens"])[:, None]
    torch.set_num_threads(min(8, os.cpu_count() or 1))

    def x0(xx, tt):
        with torch.no_grad():
            return m(torch.cat([torch.from_numpy(xx), content], dim=1), torch.from_numpy(tt), prompt, encoder_attention_mask=mask).sample.numpy()

    table = S.build_table("dpmsolver++", 1, order=1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = S.run_table_numpy(table, x0, xT.numpy())
        y2 = x0(xT.numpy(), np.array([500.0], dtype=np.float32))
    assert sum("slow plumbing path" in str(i.message) for i in w) == 1           # warned once, not per call
    assert m.cpu_calls == 2 and m.engine_calls == 0 and m.autograd_calls == 0 and np.isfinite(y2).all()
    assert rel_l2(y, gold["g5.dpm1_b1_plumbing.y"]) < 1e-4


# ---- C ABI ------------------------------------------------------------------------------
def test_cabi_library_exports_every_declared_symbol():
    from ns2vc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ns2vc_hip.h")).read()
    declared = set(re.findall(r"\b(ns2vc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ns2vc_hip.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.ns2vc_abi_version() == _lib.ABI_VERSION == 7


def test_no_packed_fp32_of_the_failing_form_under_outstanding_lds_reads():
    """Static guard for the gfx950 hazard behind round 3's non-deterministic GroupNorm prologue (profiles/r04_gn_prologue_rootcause.txt):
    no kernel of the library may hold a packed fp32 instruction whose LOW half takes a HIGH source dword (op_sel) at a point where
    LDS read returns of the wave can still be outstanding.  tools/isa_pk_lds_check.py disassembles the built objects (no GPU needed)."""
    import glob
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_pk_lds_check", os.path.join(ROOT, "tools", "isa_pk_lds_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    objs = sorted(glob.glob(os.path.join(ROOT, "ns2vc_amd", "lib", "obj", "*.o")))
    assert len(objs) >= 5, "build the library first (__graft_entry__.build())"
    sites = chk.check(objs)
    bad = [x for x in sites if chk.is_signature(x[3])]
    assert not bad, "\n".join(f"{o}: {f[:80]} {a}: {t}" for o, f, a, t, _ in bad)
    # the scanner itself: it must see the chain kernels' (harmless, counted) packed epilogue arithmetic, or it is blind
    assert any("rowchain" in x[1] or "ffn" in x[1] for x in sites)


def test_default_conv_kernels_use_no_scratch():
    """r6: an epilogue of the OFF option fuse_solver, compiled into every tap-sharing conv kernel behind a run-time test, had cost all 128-column tiles 26 spilled VGPRs and
    108 B of scratch per lane -- a scratch reload is an `s_waitcnt vmcnt(0)` beside hand-counted LDS-DMA -- and 0.4 % of the step (profiles/r06_ab_split_io.txt).  Epilogue
    and pair prologue are separate instantiations now (template parameters SOL, GNP = 3); this pins it: hipcc's kernel-resource-usage remarks for convts.hip must show no
    scratch and no spilled VGPR in any kernel of the default path (SOL = false).  The whole library: tools/kernel_resources.sh -> profiles/r06_kernel_resources.txt."""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    src = os.path.join(ROOT, "ns2vc_amd", "csrc")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I../../include", "-Rpass-analysis=kernel-resource-usage", "-c", "convts.hip",
                        "-o", os.devnull], cwd=src, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    name, seen, bad = None, 0, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line)
        if m and name and "conv3ts_kernel" in name:
            seen += 1
            # mangled template tail: ...ELi<GNP>ELb<KS>ELb<SOL>EE
            sol = re.search(r"ELb[01]ELb([01])EE", name)
            if sol and sol.group(1) == "0" and int(m.group(2)) != 0:
                bad.append((name, m.group(1), int(m.group(2))))
    assert seen >= 40, f"only {seen} resource lines parsed"
    assert not bad, bad


def test_host_operand_rounding_matches_ieee():
    """The weight-packing conversions (csrc/common.h f32_to_f16_bits / f32_to_bf16_bits) against numpy's IEEE binary16
    and the bit-level bf16 reference: random values over the whole exponent range, subnormals, ties, overflow, inf/nan."""
    import ctypes as C
    from ns2vc_amd import _lib
    from util import bf16_round, f16_round
    lib = _lib.load()
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-30, 18, 200000))).astype(np.float32)
    edge = np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, np.inf, -np.inf, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8, 2.99e-8, 1e-9,
                     1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 2.0 ** -14, 2.0 ** -24 * 1.5], dtype=np.float32)
    x = np.concatenate([x, edge, -edge])
    for prec, ref in ((2, f16_round), (1, bf16_round), (0, lambda a: a)):
        out = np.empty_like(x)
        assert lib.ns2vc_round_to_operand(x.ctypes.data_as(C.c_void_p), x.size, prec, out.ctypes.data_as(C.c_void_p)) == 0
        want = ref(x)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (prec, x[out.view(np.uint32) != want.view(np.uint32)][:8])
    nan = np.array([np.nan], dtype=np.float32)
    out = np.empty_like(nan)
    for prec in (1, 2):
        lib.ns2vc_round_to_operand(nan.ctypes.data_as(C.c_void_p), 1, prec, out.ctypes.data_as(C.c_void_p))
        assert np.isnan(out[0])


def test_geglu_stream_packing_layout():
    """r5: the weight tile stream of the token-stationary GEGLU kernel (csrc/geglu.hip), packed on the host (ns2vc_pack_geglu_host), against an
    independent restatement of its layout: [quarter][unit block][K tile][128 rows][8 chunks of 8], the rows of every 32-unit group in the order that
    makes a lane's 16 MFMA results 16 consecutive hidden units, 16-byte chunks XOR-swizzled by (row >> 1) & 7, constants in stream row order."""
    import ctypes as C
    from ns2vc_amd import _lib
    from util import bf16_round, f16_round
    lib = _lib.load()
    d = 384
    rng = np.random.default_rng(11)
    W = (rng.standard_normal((8 * d, d)) / np.sqrt(d)).astype(np.float32)
    b = rng.standard_normal(8 * d).astype(np.float32)
    # the MFMA result layout the permutation is made for: register r of lane half hi holds tile row (r & 3) + 8 (r >> 2) + 4 hi
    unit_of_row = lambda m: 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3)
    assert sorted(unit_of_row(m) for m in range(32)) == list(range(32))
    for hi in range(2):
        assert [unit_of_row((r & 3) + 8 * (r >> 2) + 4 * hi) for r in range(16)] == list(range(16 * hi, 16 * hi + 16))
    sr = np.arange(8 * d)                                               # stream row -> packed row
    pr = (sr >> 6) * 64 + ((sr >> 5) & 1) * 32 + np.array([unit_of_row(int(m)) for m in sr & 31])
    for prec, rnd, bits in ((2, f16_round, lambda a: a.astype(np.float16).view(np.uint16)), (1, bf16_round, lambda a: (a.view(np.uint32) >> 16).astype(np.uint16))):
        stream = np.zeros(8 * d * d, dtype=np.uint16)
        consts = np.zeros((8 * d, 2), dtype=np.float32)
        assert lib.ns2vc_pack_geglu_host(W.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), d, prec, stream.ctypes.data_as(C.c_void_p),
                                         consts.ctypes.data_as(C.c_void_p)) == 0
        Wr = rnd(W)
        tiles = stream.reshape(8 * d // 128, d // 64, 128, 8, 8)          # [unit block (all quarters)][K tile][row][chunk position][element]
        want = np.empty_like(tiles)
        r = np.arange(128)
        for ubg in range(8 * d // 128):
            rows = Wr[pr[ubg * 128:(ubg + 1) * 128]]                      # [128][d]
            for kt in range(d // 64):
                blk = rows[:, 64 * kt:64 * kt + 64].reshape(128, 8, 8)   # [row][logical chunk][element]
                for pos in range(8):
                    want[ubg, kt, :, pos, :] = bits(np.ascontiguousarray(blk[r, pos ^ ((r >> 1) & 7), :]))
        assert np.array_equal(tiles, want), prec
        assert np.array_equal(consts[:, 1], b[pr])
        assert np.allclose(consts[:, 0], Wr[pr].astype(np.float64).sum(1), rtol=0, atol=1e-6)
    assert lib.ns2vc_pack_geglu_host(W.ctypes.data_as(C.c_void_p), None, 256, 2, stream.ctypes.data_as(C.c_void_p), consts.ctypes.data_as(C.c_void_p)) != 0


def test_no_cpu_fallback_engine_fails_loudly_without_gpu():
    from ns2vc_amd import engine
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(engine.Ns2vcError, match="no ROCm-capable device|kernel attribute setup failed"):
        engine.Engine()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under ns2vc_amd/ or unet1d/ may import it."""
    for pkg in ("ns2vc_amd", "unet1d"):
        for fn in os.listdir(os.path.join(ROOT, pkg)):
            if fn.endswith(".py"):
                src = open(os.path.join(ROOT, pkg, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{pkg}/{fn} imports oracle"


# ---- data-parallel sharding over gloo, world size 2 ------------------------------------------
def test_shard_ranges():
    from ns2vc_amd.dist import shard_range, shard_sizes
    assert [shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    assert shard_sizes(7, 3) == [3, 2, 2]
    cover = [i for r in range(3) for i in range(*shard_range(7, r, 3))]
    assert cover == list(range(7))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ns2vc_amd.dist import shard_range, gather_latents
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 5                                   # uneven: shards of 3 and 2
full = torch.arange(n * 2 * 3, dtype=torch.float32).reshape(n, 2, 3)
lo, hi = shard_range(n, rank, world)
out = gather_latents(full[lo:hi] * 1.0, n)
assert torch.equal(out, full), (rank, out)
one = full[:1]                          # fewer items than ranks: rank 1's shard is EMPTY but it still joins the collective
lo, hi = shard_range(1, rank, world)
assert (hi - lo) == (1 if rank == 0 else 0)
out = gather_latents(one[lo:hi] * 1.0, 1)
assert torch.equal(out, one), (rank, out)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gather_latents_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


@pytest.mark.parametrize("n", [2, 8])
def test_bench_harness_dry_run_multi_rank(n, tmp_path):
    """`bench.py --gpus N`'s spawn / barrier / all-gather / per-rank reduction / strong_scaling / JSON path had never
    executed with more than one rank (no multi-GPU box so far).  `--dry-dist N` runs exactly that path on CPU: N self-spawned ranks on
    gloo, the engine replaced by a sleep + deterministic fill (every rank checks the gathered latents).  r6: the stdout line is the compact
    one (< 4 KB); with --full it is printed before the long legs and again as the last line, the long legs land in the detail file."""
    detail = tmp_path / "detail.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-dist", str(n), "--steps", "3", "--warmup", "1", "--reps", "2",
                        "--batch", "2", "--seconds", "1", "--strong-batch", "13", "--full", "--detail-json", str(detail)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and lines[0] == lines[1]            # the line, before the long legs and again last -- from rank 0 only
    assert r.stdout.rstrip().splitlines()[-1] == lines[-1] and len(lines[-1]) < 4096
    d = json.loads(lines[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "parity", "launches_per_step", "rccl_ranks"):
        assert key in d, key
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["steps"] == 3 and d["scaling"] == "weak" and d["spawned_by"] == "bench.py"
    assert len(d["per_rank_ms_per_step"]) == n and d["config"]["global_batch"] == 2 * n and d["config"]["parallelism"] == f"dp{n}"
    assert abs(d["value"] - n * 3 / (d["ms_per_step"] * 3e-3)) < 1e-4 * d["value"]          # whole-job aggregate over the ranks
    assert d["all_gather_bytes"] == 2 * n * 100 * 94 * 4 and d["all_gather_ms"] >= 0.0
    assert d["dry_dist"]["backend"] == "gloo" and d["roofline"] is None and d["cpu_baseline"] is None
    full = json.load(open(detail))
    st = full["strong_scaling"]
    assert st["global_batch"] == 13 and st["n_gpus"] == n and st["scaling"] == "strong" and st["per_rank_batch"] == (13 + n - 1) // n
    assert full["value"] == pytest.approx(d["value"], rel=1e-5)


def test_bench_default_line_is_compact_and_single():
    """the default run prints exactly ONE stdout line, valid JSON under 4 KB, carrying `roofline` and `cpu_baseline`
    (null in the GPU-less rehearsal).  `compact_line` is also exercised on a full-size record (the r5 line was 21.5 KB and went unparsed)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-dist", "1", "--steps", "3", "--warmup", "1", "--reps", "2",
                        "--batch", "2", "--seconds", "1", "--detail-json", ""], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < 4096
    d = json.loads(out[0])
    assert "roofline" in d and "cpu_baseline" in d and "parity" in d and d["n_gpus"] == 1 and d["rccl_ranks"] == 0
    import bench
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_record_r06.json")))     # a real unabridged record of the GPU box
    line = json.dumps(bench.compact_line(rec))
    assert len(line) < 4096
    c = json.loads(line)
    assert c["roofline"]["bound"] == "mfma" and 0 < c["roofline"]["frac"] < 1 and len(c["roofline"]["kernel"]) <= 120
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0 and c["parity"]["tolerance"] == 1e-3
    assert c["value"] == pytest.approx(c["n_gpus"] * 1e3 / c["ms_per_step"], rel=1e-4)


def test_repeat_expand_matches_reference_golden():
    """ns2vc_amd.audio.repeat_expand_2d (index map + one gather) against outputs of the reference's own utils.repeat_expand_2d
    (utils.py:482-496): bit-identical, including non-integer ratios, target shorter than source, and batched input."""
    import json
    from ns2vc_amd.audio import repeat_expand_2d
    from ns2vc_amd.weights import hash_normal
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v3.npz"))
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v3_report.json")))["cases"]
    assert len(cases) >= 9
    for h, s, t in cases:
        x = torch.from_numpy(hash_normal(f"g11.{h}.{s}.{t}", (h, s)))
        y = repeat_expand_2d(x, t)
        assert y.dtype == torch.float32 and np.array_equal(y.numpy(), g[f"g11.{h}_{s}_{t}.y"]), (h, s, t)
        assert torch.equal(repeat_expand_2d(torch.stack([x, 2 * x]), t)[1], 2 * y)


def test_log_mel_against_a_direct_dft():
    """ns2vc_amd.audio.log_mel (torch.stft + HTK filter bank; the reference uses torchaudio.transforms.MelSpectrogram, absent here:
    parity unpinned) against an independent numpy restatement of the published algorithm -- explicit reflect padding, framing,
    periodic Hann window, rfft magnitude, triangular HTK filters -- plus shape and a pure-tone sanity check."""
    from ns2vc_amd.audio import log_mel, mel_filterbank
    sr, n_fft, hop, n_mels = 24000, 1024, 256, 100
    rng = np.random.default_rng(3)
    t = np.arange(6000) / sr
    wav = (0.5 * np.sin(2 * np.pi * 1000.0 * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    got = log_mel(torch.from_numpy(wav)[None]).numpy()[0]
    assert got.shape == (n_mels, 1 + wav.size // hop)
    # ---- numpy restatement
    x = np.pad(wav.astype(np.float64), n_fft // 2, mode="reflect")
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)                       # periodic Hann
    frames = np.stack([x[i * hop:i * hop + n_fft] * win for i in range(1 + wav.size // hop)], axis=1)
    mag = np.abs(np.fft.rfft(frames, axis=0))                                            # (513, frames)
    hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    f_pts = hz(np.linspace(mel(0.0), mel(sr / 2), n_mels + 2))
    freqs = np.linspace(0, sr // 2, n_fft // 2 + 1)
    fb = np.zeros((n_fft // 2 + 1, n_mels))
    for m in range(n_mels):
        lo, ce, hi = f_pts[m], f_pts[m + 1], f_pts[m + 2]
        fb[:, m] = np.clip(np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)), 0.0, None)
    ref = np.log(np.clip(fb.T @ mag, 1e-7, None))
    assert rel_l2(mel_filterbank().numpy(), fb) < 1e-5
    assert np.abs(got - ref).max() < 2e-3                                                # log domain, fp32 stft vs fp64 dft
    peak = int(np.argmax(got.mean(axis=1)))
    assert f_pts[peak] <= 1000.0 <= f_pts[peak + 2]                                      # the 1 kHz tone sits in the right filter


def test_grouped_converter_plan_groups_equal_shapes():
    """ns2vc_amd.service.GroupedConverter.plan: segments are grouped by (latent length, prompt length), never padded, groups of at
    most max_batch, longest first, input order inside a group (host logic; the GPU behaviour is tests/test_service_gpu.py)."""
    from ns2vc_amd.service import GroupedConverter, Segment
    gc = GroupedConverter.__new__(GroupedConverter)
    gc.max_batch = 3
    lengths = [96, 130, 96, 64, 130, 96, 96, 96]
    rlens = [40, 64, 40, 64, 64, 40, 50, 40]
    segs = [Segment(torch.zeros(256, T), torch.zeros(100, L)) for T, L in zip(lengths, rlens)]
    groups = gc.plan(segs)
    assert groups == [[1, 4], [6], [0, 2, 5], [7], [3]]
    assert sorted(i for g_ in groups for i in g_) == list(range(len(segs)))
    for g_ in groups:
        assert len({(lengths[i], rlens[i]) for i in g_}) == 1 and len(g_) <= 3


def test_vocos_decoder_restatement_shapes_keys_and_inverse_stft():
    """ns2vc_amd/vocoder.py (the back end `vocos.decode`, model.py:689-691; parity UNPINNED: the vocos package and its
    checkpoint are not available offline): parameter names / shapes of the published `charactr/vocos-mel-24khz` decoder, the
    output length the reference's slicing relies on ((T - 1) * 256 samples), and the spectrum head + inverse STFT against an
    independent numpy overlap-add."""
    from ns2vc_amd.vocoder import VocosDecoder
    torch.manual_seed(0)
    m = VocosDecoder().eval()
    sd = m.state_dict()
    assert sd["backbone.embed.weight"].shape == (512, 100, 7) and sd["head.out.weight"].shape == (1026, 512) and sd["head.istft.window"].shape == (1024,)
    for i in range(8):
        assert sd[f"backbone.convnext.{i}.dwconv.weight"].shape == (512, 1, 7) and sd[f"backbone.convnext.{i}.pwconv1.weight"].shape == (1536, 512)
        assert sd[f"backbone.convnext.{i}.pwconv2.weight"].shape == (512, 1536) and sd[f"backbone.convnext.{i}.gamma"].shape == (512,)
    assert len(sd) == 4 + 8 * 9 + 2 + 2 + 1 and sum(v.numel() for k, v in sd.items() if k != "head.istft.window") == 13_531_650
    m.load_vocos_state_dict({**sd, "feature_extractor.mel_spec.mel_scale.fb": torch.zeros(513, 100)})       # encoder-side buffers are ignored
    B, T = 2, 41
    mel = torch.randn(B, 100, T)
    audio = m.decode(mel)
    assert audio.shape == (B, (T - 1) * 256) and bool(torch.isfinite(audio).all())
    # spectrum head + inverse STFT vs numpy: irfft of every frame, Hann window, overlap-add, window-envelope normalisation, centre trim
    h = torch.randn(B, T, 512)
    with torch.no_grad():
        got = m.head(h).numpy()
        o = (h @ m.head.out.weight.T + m.head.out.bias).numpy().astype(np.float64).transpose(0, 2, 1)
    mag, ph = np.minimum(np.exp(o[:, :513]), 1e2), o[:, 513:]
    spec = mag * (np.cos(ph) + 1j * np.sin(ph))
    win = np.hanning(1025)[:1024]                                   # periodic Hann = torch.hann_window(1024)
    n = 1024 + 256 * (T - 1)
    want = np.zeros((B, n)); env = np.zeros(n)
    for t in range(T):
        want[:, 256 * t: 256 * t + 1024] += np.fft.irfft(spec[:, :, t], n=1024, axis=-1) * win
        env[256 * t: 256 * t + 1024] += win ** 2
    want = (want / np.maximum(env, 1e-11))[:, 512: n - 512]
    assert got.shape == want.shape and rel_l2(got, want) < 1e-4


def test_contentvec_restatement_shapes_names_and_pieces():
    """ns2vc_amd/contentvec.py restates the HuBERT-base / ContentVec encoder the reference loads through fairseq (utils.py:209-236;
    parity unpinned: neither fairseq nor the checkpoint exists offline).  What CAN be pinned here: the frame arithmetic (hop 320,
    receptive field 400 -> 50 Hz), the published parameter count and fairseq's key names, the weight-norm reconstruction of
    pos_conv against torch's own weight_norm, the attention against torch's multi_head_attention_forward, and the output shape."""
    import torch
    import torch.nn.functional as F
    from ns2vc_amd.contentvec import ContentVec, PosConv, SelfAttention
    torch.manual_seed(0)
    m = ContentVec().eval()
    assert ContentVec.frames_for(16000) == 49 and ContentVec.frames_for(400) == 1 and ContentVec.frames_for(160000) == 499
    sd = m.state_dict()
    assert sum(v.numel() for v in sd.values()) == 94_567_808
    for k, shape in {"feature_extractor.conv_layers.0.0.weight": (512, 1, 10), "feature_extractor.conv_layers.0.2.weight": (512,),
                     "feature_extractor.conv_layers.6.0.weight": (512, 512, 2), "layer_norm.weight": (512,), "post_extract_proj.weight": (768, 512),
                     "encoder.pos_conv.0.weight_g": (1, 1, 128), "encoder.pos_conv.0.weight_v": (768, 48, 128), "encoder.pos_conv.0.bias": (768,),
                     "encoder.layers.11.self_attn.q_proj.weight": (768, 768), "encoder.layers.0.fc1.weight": (3072, 768),
                     "encoder.layers.5.final_layer_norm.bias": (768,), "encoder.layer_norm.weight": (768,), "final_proj.weight": (256, 768)}.items():
        assert tuple(sd[k].shape) == shape, k
    norms = [k for k in sd if k.startswith("feature_extractor.conv_layers.") and k.split(".")[3] == "2"]
    assert norms == ["feature_extractor.conv_layers.0.2.weight", "feature_extractor.conv_layers.0.2.bias"]      # only the first conv block carries a norm
    # a fairseq-style state dict (with the pre-training leftovers) loads; a missing key is an error
    full = dict(sd, mask_emb=torch.zeros(768), label_embs_concat=torch.zeros(504, 256))
    m.load_fairseq_state_dict(full)
    with pytest.raises(RuntimeError):
        m.load_fairseq_state_dict({k: v for k, v in full.items() if k != "final_proj.bias"})
    # weight normalisation over dim 2, as fairseq builds it
    pc = PosConv()
    ref = torch.nn.utils.weight_norm(torch.nn.Conv1d(768, 768, 128, padding=64, groups=16), name="weight", dim=2)
    with torch.no_grad():
        ref.weight_g.copy_(torch.rand_like(ref.weight_g) + 0.5); ref.weight_v.copy_(torch.randn_like(ref.weight_v))
        pc.weight_g.copy_(ref.weight_g); pc.weight_v.copy_(ref.weight_v); pc.bias.copy_(ref.bias)
        x = torch.randn(2, 768, 37)
        assert torch.allclose(pc(x), F.gelu(ref(x)[:, :, :-1]), atol=1e-5)
    # the attention block against torch's reference implementation with separate projection weights
    sa = SelfAttention(768, 12).eval()
    with torch.no_grad():
        x = torch.randn(2, 29, 768)
        want, _ = F.multi_head_attention_forward(
            x.transpose(0, 1), x.transpose(0, 1), x.transpose(0, 1), 768, 12, None, torch.cat([sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias]),
            None, None, False, 0.0, sa.out_proj.weight, sa.out_proj.bias, training=False, need_weights=False, use_separate_proj_weight=True,
            q_proj_weight=sa.q_proj.weight, k_proj_weight=sa.k_proj.weight, v_proj_weight=sa.v_proj.weight)
        assert torch.allclose(sa(x), want.transpose(0, 1), atol=2e-5)
        # end to end: (B, samples) -> (B, 256, frames), batched == one by one
        wav = torch.randn(2, 8000) * 0.1
        y = m.extract(wav)
        assert y.shape == (2, 256, ContentVec.frames_for(8000)) and torch.isfinite(y).all()
        assert torch.allclose(m.extract(wav[1]), y[1:2], atol=1e-4)
        # the converter's unit of work straight from audio: 0.5 s at 24 kHz = 12000 samples -> 46 latent frames
        from ns2vc_amd.service import segment_from_audio
        seg = segment_from_audio(m, wav[0], 12000, torch.zeros(100, 40), tag="a")
        assert seg.content.shape == (256, 12000 // 256) and seg.refer.shape == (100, 40) and seg.tag == "a"


THIRDPARTY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_thirdparty.npz")
_needs_thirdparty = pytest.mark.skipif(not os.path.exists(THIRDPARTY), reason="tests/golden/golden_thirdparty.npz not generated yet: run "
                                       "tests/golden/make_golden_thirdparty.py where vocos / torchaudio / fairseq exist (f2 / f4 stay parity-unpinned until then)")


@_needs_thirdparty
def test_thirdparty_goldens_log_mel():
    """f2, prompt side: ns2vc_amd.audio.log_mel against torchaudio's MelSpectrogram + log-clip as the reference calls it (infer_tool.py:170-182)"""
    from ns2vc_amd.audio import log_mel
    g = np.load(THIRDPARTY)
    if "mel.log_mel" not in g.files:
        pytest.skip("the fixture holds no mel.* block")
    y = log_mel(torch.from_numpy(g["mel.wav24k"])).numpy()
    assert y.shape == g["mel.log_mel"].shape and rel_l2(y, g["mel.log_mel"]) < 1e-5


@_needs_thirdparty
def test_thirdparty_goldens_vocos_decode():
    """f2, back end: ns2vc_amd.vocoder.VocosDecoder.decode against vocos.decode (model.py:689-691); needs the vocoder's state dict (NS2VC_VOCOS_STATE)"""
    from ns2vc_amd.vocoder import VocosDecoder
    g = np.load(THIRDPARTY)
    path = os.environ.get("NS2VC_VOCOS_STATE", "")
    if "vocos.audio" not in g.files or not os.path.exists(path):
        pytest.skip("no vocos.* block in the fixture, or NS2VC_VOCOS_STATE does not point at the vocoder's state dict")
    dec = VocosDecoder().eval()
    dec.load_vocos_state_dict(torch.load(path, map_location="cpu"))
    y = dec.decode(torch.from_numpy(g["vocos.mel"])).numpy()
    assert y.shape == g["vocos.audio"].shape and rel_l2(y, g["vocos.audio"]) < 1e-4


@_needs_thirdparty
def test_thirdparty_goldens_contentvec():
    """f4: ns2vc_amd.contentvec.ContentVec.extract against fairseq's extract_features(output_layer=12) + final_proj (utils.py:221-236); needs the checkpoint (NS2VC_HUBERT_CKPT)"""
    from ns2vc_amd.contentvec import ContentVec
    g = np.load(THIRDPARTY)
    path = os.environ.get("NS2VC_HUBERT_CKPT", "")
    if "hubert.content" not in g.files or not os.path.exists(path):
        pytest.skip("no hubert.* block in the fixture, or NS2VC_HUBERT_CKPT does not point at checkpoint_best_legacy_500.pt")
    cv = ContentVec().eval()
    cv.load_fairseq_state_dict(torch.load(path, map_location="cpu")["model"])
    y = cv.extract(torch.from_numpy(g["hubert.wav16k"])).numpy()
    assert y.shape == g["hubert.content"].shape and rel_l2(y, g["hubert.content"]) < 1e-4
