"""fp16 forward error vs the oracle for several procedural weight seeds and timesteps (is 8.2e-4 a property of seed 0?)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import rel_l2
from ns2vc_amd.engine import Engine
from ns2vc_amd.spec import UNetConfig
from ns2vc_amd.weights import hash_normal, procedural_state_dict
from oracle import unet_ref
torch.set_num_threads(16)
B, T, Lp = 4, 938, 469
for seed in (0, 1, 2, 3):
    W = procedural_state_dict(UNetConfig(), seed)
    P = {k: torch.from_numpy(v) for k, v in W.items()}
    x, c, p = (hash_normal(f"sp{seed}.{n}", s) for n, s in (("x", (B, 100, T)), ("c", (B, 256, T)), ("p", (B, Lp, 256))))
    mask = np.ones((B, Lp), dtype=bool); mask[1, 300:] = False
    t = np.array([40.0, 350.5, 700.0, 960.0], dtype=np.float32)
    ref = unet_ref.denoiser(P, UNetConfig(), torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(p), torch.from_numpy(mask), torch.from_numpy(t)).numpy()
    out = {}
    for prec in ("fp16", "fp32"):
        e = Engine(precision=prec); e.load_state_dict(W); e.prepare(B, T, Lp)
        d = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in (x, c, p, mask.astype(np.uint8), t)]
        y = torch.empty_like(d[0])
        e.set_condition(d[1], d[2], d[3]); e.forward(d[0], d[4], y); torch.cuda.synchronize()
        yy = y.cpu().numpy()
        out[prec] = (rel_l2(yy, ref), [rel_l2(yy[i], ref[i]) for i in range(B)])
        e.close()
    print(f"seed {seed}: fp16 {out['fp16'][0]:.3e} (per item {[f'{v:.2e}' for v in out['fp16'][1]]})  fp32 {out['fp32'][0]:.3e}", flush=True)
