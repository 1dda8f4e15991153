#!/usr/bin/env python3
"""Run-to-run and graph-vs-eager determinism of the sampling loop at the bench shape (GPU box).
    python tools/determinism_probe.py [--steps 4] [--batch 32]
Prints, for the engine's current plan: graph loop twice, eager loop twice, all pairs compared bitwise."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bench_inputs
from ns2vc_amd import engine as E
from ns2vc_amd.spec import UNetConfig, frames_for_seconds
from ns2vc_amd.weights import procedural_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--more", type=int, default=0, help="additional captured-loop runs (then only the comparisons with the first run are printed)")
ap.add_argument("--forwards", type=int, default=0, help="also N single forwards (per-item timesteps, partly masked prompt), each compared with the first")
a = ap.parse_args()
cfg = UNetConfig()
B, T, Lp = a.batch, frames_for_seconds(a.seconds), 469
dev = torch.device("cuda", 0)
eng = E.Engine(cfg, precision="fp16")
eng.load_state_dict(procedural_state_dict(cfg, 0))
eng.prepare(B, T, Lp)
eng.load_sampler("unipc", a.steps, order=2)
n_np, c_np, p_np = bench_inputs("det", B, T, Lp)
c, p_, n = (torch.from_numpy(v).to(dev) for v in (c_np, p_np, n_np))
mask = torch.ones((B, Lp), dtype=torch.uint8, device=dev)
stream = torch.cuda.Stream(device=dev)
outs = {}
with torch.cuda.stream(stream):
    runs = [("graph1", True), ("graph2", True), ("eager1", False), ("eager2", False), ("graph3", True)] + [(f"graph{i}", True) for i in range(4, 4 + a.more)]
    for tag, g in runs:
        x = n.clone()
        eng.set_condition(c, p_, mask, stream=stream)
        eng.sample(x, use_graph=g, stream=stream)
        stream.synchronize()
        outs[tag] = x.clone()
keys = list(outs)
for i in range(len(keys)):
    for j in range(i + 1, len(keys)):
        if i > 0 and a.more:
            continue
        d = (outs[keys[i]].double() - outs[keys[j]].double())
        print(f"{keys[i]} vs {keys[j]}: equal={bool(torch.equal(outs[keys[i]], outs[keys[j]]))} rel={float(d.norm() / outs[keys[i]].double().norm()):.3e} "
              f"differing elements={int((d != 0).sum())}")
if a.forwards:
    m2 = mask.clone(); m2[B - 1, Lp // 2:] = 0
    t = torch.linspace(40.0, 960.0, B, device=dev)
    first, nbad = None, 0
    with torch.cuda.stream(stream):
        for i in range(a.forwards):
            eng.set_condition(c, p_, m2, stream=stream)
            y = torch.empty_like(n)
            eng.forward(n, t, y, stream=stream)
            stream.synchronize()
            if first is None:
                first = y.clone()
            elif not torch.equal(y, first):
                nbad += 1
                d = (y.double() - first.double())
                print(f"forward {i}: differs from forward 0: rel {float(d.norm() / first.double().norm()):.3e}, {int((d != 0).sum())} elements")
    print(f"{a.forwards} forwards: {nbad} differ from the first")
print("launches", eng.launches()[0])
