#!/usr/bin/env python3
"""r6: where does the fp16 engine's forward error (8.2e-4 of the 1e-3 bar at the bench shape) come from?  CPU emulation on the
oracle (test infrastructure; no GPU): the fp16 engine's operand rounding -- activations AND weights of every MFMA, the probabilities of every
attention -- is injected into the oracle's conv / linear / attention calls at ONE site class at a time (everything else exact fp32), then at all of
them (the emulation r3 validated against the MI355X: 8.3e-4 predicted, 8.2e-4 measured).  If the error were concentrated, giving the two or three worst
sites fp32 (or hi + lo fp16) operands would buy margin for byte diets; if it is diffuse, nothing short of a wider operand type everywhere does.

    python tools/error_budget.py [--frames 938] [--batch 2]        -> profiles/r06_error_budget.txt

Site classes (by the weight a call multiplies with; attention products by the block they sit in):
  conv_in, res.conv1, res.conv2, res.shortcut, down/up-sample convs, proj_in, attn1.qkv, attn1.QK^T+PV (operands q, k, v and the probabilities),
  attn1.to_out, attn2.to_q, attn2.k|v (hoisted per utterance), attn2.QK^T+PV, attn2.to_out, ff.geglu (ff.net.0), ff.out (ff.net.2), proj_out, conv_out,
  time (time_embedding + time_emb_proj + add_embedding linears).
The squared errors of independent sites add: `rss` = root of the sum of the single-site squares, to compare with `all`."""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ns2vc_amd.spec import UNetConfig                                  # noqa: E402
from ns2vc_amd.weights import hash_normal, procedural_state_dict      # noqa: E402
from oracle import unet_ref                                            # noqa: E402

SITES = ["conv_in", "res.conv1", "res.conv2", "res.shortcut", "resample", "proj_in", "attn1.qkv", "attn1.sdpa", "attn1.to_out", "attn2.to_q", "attn2.kv",
         "attn2.sdpa", "attn2.to_out", "ff.geglu", "ff.out", "proj_out", "conv_out", "time"]


def r16(x):
    return x.to(torch.float16).to(torch.float32)


def classify(name: str) -> str:
    if name.startswith("conv_in"):
        return "conv_in"
    if name.startswith("conv_out"):
        return "conv_out"
    if ".conv_shortcut." in name:
        return "res.shortcut"
    if ".resnets." in name and ".conv1." in name:
        return "res.conv1"
    if ".resnets." in name and ".conv2." in name:
        return "res.conv2"
    if "samplers." in name:
        return "resample"
    if ".proj_in." in name:
        return "proj_in"
    if ".proj_out." in name:
        return "proj_out"
    if ".attn1.to_out" in name:
        return "attn1.to_out"
    if ".attn2.to_out" in name:
        return "attn2.to_out"
    if ".attn1.to_" in name:
        return "attn1.qkv"
    if ".attn2.to_q" in name:
        return "attn2.to_q"
    if ".attn2.to_k" in name or ".attn2.to_v" in name:
        return "attn2.kv"
    if ".ff.net.0." in name:
        return "ff.geglu"
    if ".ff.net.2." in name:
        return "ff.out"
    return "time"                                                       # time_embedding.*, *.time_emb_proj.*, add_embedding.*


class Proxy:
    """torch.nn.functional with the fp16 engine's rounding points switched on for the site classes in `on`"""

    def __init__(self, names, on):
        self.names, self.on, self.attn = names, on, "attn1"

    def __getattr__(self, k):
        return getattr(TF, k)

    def _site(self, w):
        return classify(self.names.get(id(w), "time"))

    def conv1d(self, x, w, b=None, **kw):
        return TF.conv1d(r16(x), r16(w), b, **kw) if self._site(w) in self.on else TF.conv1d(x, w, b, **kw)

    def linear(self, x, w, b=None):
        return TF.linear(r16(x), r16(w), b) if self._site(w) in self.on else TF.linear(x, w, b)

    def scaled_dot_product_attention(self, q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
        if self.attn + ".sdpa" not in self.on:
            return TF.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
        q, k, v = r16(q), r16(k), r16(v)
        s = (q @ k.transpose(-1, -2)) / (q.shape[-1] ** 0.5)
        if attn_mask is not None:
            s = s + attn_mask
        p = torch.softmax(s, dim=-1)
        pm = p.amax(dim=-1, keepdim=True)
        pr = r16(p / pm)                                               # probabilities rounded relative to the row maximum, denominator from the rounded ones
        return (pr @ v) / pr.sum(-1, keepdim=True)


def run(P, names, cfg, on, x, content, prompt, t):
    proxy = Proxy(names, set(on))
    saved_F, saved_att = unet_ref.F, unet_ref._attention

    def attention(P_, pre, heads, xx, ctx, bias):
        proxy.attn = "attn2" if pre.endswith("attn2") else "attn1"
        return saved_att(P_, pre, heads, xx, ctx, bias)
    unet_ref.F, unet_ref._attention = proxy, attention
    try:
        return unet_ref.denoiser(P, cfg, x, content, prompt, None, t)
    finally:
        unet_ref.F, unet_ref._attention = saved_F, saved_att


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=938)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--prompt", type=int, default=469)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_error_budget.txt"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    cfg = UNetConfig()
    B, T, Lp = a.batch, a.frames, a.prompt
    rows = {s: [] for s in SITES + ["all"]}
    t0 = time.time()
    cols = []
    for seed in (0,):
        P = {k: torch.from_numpy(v) for k, v in procedural_state_dict(cfg, seed).items()}
        names = {id(v): k for k, v in P.items()}
        x, content, prompt = (torch.from_numpy(hash_normal(f"budget{seed}.{n}", s)) for n, s in (("x", (B, 100, T)), ("c", (B, 256, T)), ("p", (B, Lp, 256))))
        for tval in (900.0, 300.0, 40.0):
            cols.append(f"seed{seed} t={tval:g}")
            t = torch.full((B,), tval)
            ref = run(P, names, cfg, (), x, content, prompt, t).double()
            for site in SITES:
                y = run(P, names, cfg, (site,), x, content, prompt, t).double()
                rows[site].append(float((y - ref).norm() / ref.norm()))
            y = run(P, names, cfg, SITES, x, content, prompt, t).double()
            rows["all"].append(float((y - ref).norm() / ref.norm()))
            print(f"... {cols[-1]} done after {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    n = len(cols)
    rss = [sum(rows[s][i] ** 2 for s in SITES) ** 0.5 for i in range(n)]
    lines = [f"# {__doc__.splitlines()[0]}",
             f"# oracle emulation (tools/error_budget.py), batch {B}, T = {T}, Lp = {Lp}: rel-L2 of the predicted latent vs the exact fp32 oracle with fp16 operand rounding at ONE site class",
             f"# columns: {', '.join(cols)} | share = the site's square over the sum of squares (mean over the columns)", ""]
    tot = [r * r for r in rss]
    order = sorted(SITES, key=lambda s: -sum(rows[s][i] ** 2 / tot[i] for i in range(n)))
    lines.append(f"{'site class':14s} " + " ".join(f"{c:>14s}" for c in cols) + f" {'share':>8s}")
    for s in order:
        share = sum(rows[s][i] ** 2 / tot[i] for i in range(n)) / n
        lines.append(f"{s:14s} " + " ".join(f"{v:14.2e}" for v in rows[s]) + f" {100 * share:7.1f}%")
    lines.append(f"{'rss of sites':14s} " + " ".join(f"{v:14.2e}" for v in rss))
    lines.append(f"{'all sites':14s} " + " ".join(f"{v:14.2e}" for v in rows['all']))
    # what exact operands at the k worst sites would leave
    lines.append("")
    for k in (1, 2, 3, 5):
        left = [(sum(rows[s][i] ** 2 for s in order[k:])) ** 0.5 for i in range(n)]
        lines.append(f"# exact operands at the {k} worst site classes ({', '.join(order[:k])}) would leave " + " ".join(f"{v:.2e}" for v in left))
    txt = "\n".join(lines) + "\n"
    with open(a.out, "w") as f:
        f.write(txt)
    print(txt)


if __name__ == "__main__":
    main()
