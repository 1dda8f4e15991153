"""bring-up probe of the GroupNorm-in-producer option at the bench shape: eager forwards, a graph loop, per-launch times of the
conv1 launches, with the fault count after each phase"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from ns2vc_amd import engine as E
from ns2vc_amd.spec import UNetConfig
from ns2vc_amd.weights import procedural_state_dict, hash_normal

B, T, Lp = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 938, 469)))
cfg = UNetConfig()
dev = torch.device("cuda", 0)
eng = E.Engine(cfg, precision="fp16")
eng.load_state_dict(procedural_state_dict(cfg, 0))
eng.prepare(B, T, Lp)
eng.load_sampler("unipc", 20, order=2)
noise = torch.from_numpy(hash_normal("p.n", (B, cfg.latent_channels, T))).to(dev)
content = torch.from_numpy(hash_normal("p.c", (B, cfg.content_channels, T))).to(dev)
prompt = torch.from_numpy(hash_normal("p.p", (B, Lp, cfg.cross_attention_dim))).to(dev)
mask = torch.ones((B, Lp), dtype=torch.uint8, device=dev)
t = torch.full((B,), 500.0, device=dev)
out = torch.empty_like(noise)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    eng.set_condition(content, prompt, mask, stream=st)
    for i in range(4):
        t0 = time.perf_counter()
        eng.forward(noise, t, out, stream=st)
        st.synchronize()
        print(f"eager forward {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms  faults {eng.sync_faults()}", flush=True)
    for g in (True, False, True):
        x = noise.clone()
        t0 = time.perf_counter()
        eng.sample(x, use_graph=g, stream=st)
        st.synchronize()
        print(f"sample graph={g}: {(time.perf_counter() - t0) * 1e3:.2f} ms  faults {eng.sync_faults()}  sum {float(x.double().sum()):.6f}", flush=True)
    ops = eng.op_info(0)
    for i in range(6):
        t0 = time.perf_counter()
        ms = eng.profile_forward(reps=8, stream=st)
        print(f"profile_forward {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms  faults {eng.sync_faults()}", flush=True)
    for (name, kind, fl, by), m in zip(ops, ms):
        if name.endswith(".conv1") or name.endswith(".norm2.gn_apply"):
            print(f"  {name:40s} {m * 1e3:8.1f} us")
