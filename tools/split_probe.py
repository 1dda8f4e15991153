"""Does running the batch as S independent sub-batches on S streams (S captured graphs in flight) beat one graph?
The denoiser's kernels are latency-bound at B=32 (MFMA busy 10-25 %), and the utterances of a batch are independent."""
import os, sys, time, statistics
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ns2vc_amd import engine as E
from ns2vc_amd.spec import UNetConfig
from ns2vc_amd.weights import procedural_state_dict
dev = torch.device("cuda", 0)
B, T, Lp, K = 32, 938, 469, 20
cfg = UNetConfig(); W = procedural_state_dict(cfg, 0)
prec = os.environ.get("PREC", "fp16")
g = torch.Generator(device=dev).manual_seed(1)
content = torch.randn((B, 256, T), device=dev, generator=g); prompt = torch.randn((B, Lp, 256), device=dev, generator=g)
noise = torch.randn((B, 100, T), device=dev, generator=g); mask = torch.ones((B, Lp), dtype=torch.uint8, device=dev)
results = {}
for S in [int(v) for v in (sys.argv[1:] or ["1", "2", "4"])]:
    b = B // S
    engs, streams, xs = [], [], []
    for i in range(S):
        e = E.Engine(cfg, precision=prec); e.load_state_dict(W); e.prepare(b, T, Lp); e.load_sampler("unipc", K, order=2)
        engs.append(e); streams.append(torch.cuda.Stream(device=dev)); xs.append(torch.empty((b, 100, T), device=dev))
    def job():
        for i, (e, s) in enumerate(zip(engs, streams)):
            sl = slice(i * b, (i + 1) * b)
            with torch.cuda.stream(s):
                xs[i].copy_(noise[sl])
                e.set_condition(content[sl], prompt[sl], mask[sl], stream=s)
                e.sample(xs[i], use_graph=True, stream=s)
    for _ in range(2): job()
    torch.cuda.synchronize()
    walls = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); job(); torch.cuda.synchronize(); walls.append(time.perf_counter() - t0)
    out = torch.cat(xs).clone()
    results[S] = out
    ms = statistics.median(walls) * 1e3
    print(f"S={S} sub-batches of {b}: {ms:.2f} ms per {K}-step job = {ms / K:.3f} ms/step (min {min(walls) * 1e3 / K:.3f}); "
          f"vs S=1 output rel {float((out - results[min(results)]).norm() / results[min(results)].norm()):.2e}", flush=True)
    for e in engs: e.close()
