"""does the denoiser's output depend on what a previous kernel left in LDS / registers?  (ns2vc_debug_poison)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import rel_l2
from ns2vc_amd import _lib
from ns2vc_amd.pipeline import Denoiser
from ns2vc_amd.weights import procedural_state_dict
dev = torch.device("cuda", 0)
B, T, Lp = [int(v) for v in os.environ.get("SHAPE", "32,938,469").split(",")]
steps = 6
g = torch.Generator(device=dev).manual_seed(5)
content = torch.randn((B, 256, T), device=dev, generator=g); prompt = torch.randn((B, Lp, 256), device=dev, generator=g)
mask = torch.ones((B, Lp), dtype=torch.bool, device=dev); mask[-1, Lp // 2:] = False
noise = torch.randn((B, 100, T), device=dev, generator=g)
t = torch.full((B,), 500.0, device=dev)
r = lambda x, y: rel_l2(x.cpu().numpy(), y.cpu().numpy())
lib = _lib.load()
def poison(pat, nbytes=160 * 1024):
    assert lib.ns2vc_debug_poison(pat, nbytes, None) == 0
    torch.cuda.synchronize()
for prec in sys.argv[1:] or ["fp16", "bf16", "fp32"]:
    den = Denoiser(procedural_state_dict(seed=0), precision=prec, ln_guard=None)
    F = den.denoise(noise, t, content, prompt, mask)
    A = den.sample(content, prompt, mask, noise, solver="unipc", steps=steps)
    print(prec, "steady repeat:", r(den.denoise(noise, t, content, prompt, mask), F), r(den.sample(content, prompt, mask, noise, solver="unipc", steps=steps), A), flush=True)
    for pat in (0x7FC00000, 0x7E007E00, 0xFFFFFFFF, 0x7F7F7F7F, 0x477F477F, 0):
        poison(pat); e_f = r(den.denoise(noise, t, content, prompt, mask), F)
        poison(pat); e_g = r(den.sample(content, prompt, mask, noise, solver="unipc", steps=steps), A)
        poison(pat); e_e = r(den.sample(content, prompt, mask, noise, solver="unipc", steps=steps, use_graph=False), A)
        print(prec, f"poison {pat:#010x}: forward {e_f:.3e}  graph loop {e_g:.3e}  eager loop {e_e:.3e}", flush=True)
