"""call-order dependence of the sampler loop: F = plain forward, G = captured-graph loop, E = eager loop"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import rel_l2
from ns2vc_amd.pipeline import Denoiser
from ns2vc_amd.weights import procedural_state_dict
dev = torch.device("cuda", 0)
B, T, Lp = [int(v) for v in os.environ.get("SHAPE", "32,938,469").split(",")]
steps = 6
g = torch.Generator(device=dev).manual_seed(5)
content = torch.randn((B, 256, T), device=dev, generator=g); prompt = torch.randn((B, Lp, 256), device=dev, generator=g)
mask = torch.ones((B, Lp), dtype=torch.bool, device=dev); mask[-1, Lp // 2:] = False
if os.environ.get("NOMASK"): mask = None
noise = torch.randn((B, 100, T), device=dev, generator=g)
t = torch.full((B,), 500.0, device=dev)
r = lambda x, y: rel_l2(x.cpu().numpy(), y.cpu().numpy())
den = Denoiser(procedural_state_dict(seed=0), precision=os.environ.get("PREC", "fp16"), ln_guard=None)
ref = {}
def run(k):
    if k == "F": y = den.denoise(noise, t, content, prompt, mask)
    else: y = den.sample(content, prompt, mask, noise, solver="unipc", steps=steps, use_graph=(k == "G"))
    kk = "F" if k == "F" else "S"
    if kk not in ref: ref[kk] = y
    return r(y, ref[kk])
for seq in sys.argv[1:]:
    print(seq, " ".join(f"{k}:{run(k):.3e}" for k in seq), flush=True)
