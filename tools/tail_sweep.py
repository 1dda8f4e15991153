#!/usr/bin/env python3
"""Sampled-latent error of the 16-bit loop vs the fp32 loop as a function of how many LAST evaluations run on the fp32 engine
(GPU box).   python tools/tail_sweep.py [--solver dpmsolver++] [--steps 50] [--batch 8] [--seconds 10]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bench_inputs
from ns2vc_amd import engine as E
from ns2vc_amd.spec import UNetConfig, frames_for_seconds
from ns2vc_amd.weights import procedural_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--solver", default="dpmsolver++")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
cfg = UNetConfig()
B, T, Lp = a.batch, frames_for_seconds(a.seconds), 469
dev = torch.device("cuda", 0)
W = procedural_state_dict(cfg, 0)
engs = {}
for p in (a.precision, "fp32"):
    e = E.Engine(cfg, precision=p)
    e.load_state_dict(W); e.prepare(B, T, Lp); e.load_sampler(a.solver, a.steps, order=2)
    engs[p] = e
n_np, c_np, p_np = bench_inputs("tail", B, T, Lp)
c, p_, n = (torch.from_numpy(v).to(dev) for v in (c_np, p_np, n_np))
mask = torch.ones((B, Lp), dtype=torch.uint8, device=dev)
stream = torch.cuda.Stream(device=dev)


def run(eng, tail=None, k=0):
    x = n.clone()
    with torch.cuda.stream(stream):
        eng.set_condition(c, p_, mask, stream=stream)
        if tail is not None:
            tail.set_condition(c, p_, mask, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sample(x, use_graph=True, stream=stream, tail=tail, tail_steps=k)
        stream.synchronize()
    return x, (time.perf_counter() - t0) * 1e3


ref, _ = run(engs["fp32"])
run(engs[a.precision], engs["fp32"], 1)           # warm-up (graphs)
for k in (0, 1, 2, 3, 4):
    x, _ = run(engs[a.precision], engs["fp32"] if k else None, k)
    x, ms = run(engs[a.precision], engs["fp32"] if k else None, k)
    per_item = ((x.double() - ref.double()).flatten(1).norm(dim=1) / ref.double().flatten(1).norm(dim=1))
    print(f"{a.solver}-{a.steps} B={B} T={T} {a.precision}, last {k} evaluations fp32: sampled latent vs fp32 loop {float((x.double() - ref.double()).norm() / ref.double().norm()):.3e} "
          f"(worst item {float(per_item.max()):.3e}); loop {ms:.1f} ms = {ms / a.steps:.3f} ms/step", flush=True)
