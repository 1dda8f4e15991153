#!/usr/bin/env python3
"""Experiment: does running the batch as S independent sub-batches on S streams (each with its own
captured step graph) overlap the latency-bound kernels?  python tools/split_batch_probe.py --split 2"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import engine as E
from ns2vc_amd.spec import UNetConfig
from ns2vc_amd.weights import hash_normal, procedural_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--split", type=int, default=2)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
cfg = UNetConfig()
W = procedural_state_dict(cfg, 0)
T, Lp = 938, 469
for split in sorted({1, a.split}):
    b = a.batch // split
    engs, bufs, streams = [], [], []
    for i in range(split):
        e = E.Engine(cfg, precision="bf16"); e.load_state_dict(W); e.prepare(b, T, Lp); e.load_sampler("unipc", a.steps)
        c = E.DevBuf.from_numpy(hash_normal(f"c{i}", (b, 256, T))); p = E.DevBuf.from_numpy(hash_normal(f"p{i}", (b, Lp, 256)))
        x = E.DevBuf.from_numpy(hash_normal(f"x{i}", (b, 100, T)))
        engs.append(e); bufs.append((c, p, x)); streams.append(E.Stream())
    def job():
        for e, (c, p, x), s in zip(engs, bufs, streams):
            e.set_condition(c, p, None, stream=s)
        for e, (c, p, x), s in zip(engs, bufs, streams):
            e.sample(x, use_graph=True, stream=s)
        for s in streams:
            s.sync()
    job(); job()
    t0 = time.perf_counter(); job(); dt = time.perf_counter() - t0
    print(f"split={split}: {dt*1e3/a.steps:.3f} ms/step for batch {a.batch} ({a.steps/dt:.1f} steps/s)")
