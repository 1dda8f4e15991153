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
    for tag, g in (("graph1", True), ("graph2", True), ("eager1", False), ("eager2", False), ("graph3", True)):
        x = n.clone()
        eng.set_condition(c, p_, mask, stream=stream)
        eng.sample(x, use_graph=g, stream=stream)
        stream.synchronize()
        outs[tag] = x.clone()
keys = list(outs)
for i in range(len(keys)):
    for j in range(i + 1, len(keys)):
        d = (outs[keys[i]].double() - outs[keys[j]].double())
        print(f"{keys[i]} vs {keys[j]}: equal={bool(torch.equal(outs[keys[i]], outs[keys[j]]))} rel={float(d.norm() / outs[keys[i]].double().norm()):.3e} "
              f"differing elements={int((d != 0).sum())}")
print("launches", eng.launches()[0])
