import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ns2vc_amd import engine as E
from ns2vc_amd.spec import UNetConfig
from ns2vc_amd.weights import hash_normal, procedural_state_dict
cfg = UNetConfig()
e = E.Engine(cfg, precision="bf16"); e.load_state_dict(procedural_state_dict(cfg, 0))
B, T, Lp = 32, 938, 469
e.prepare(B, T, Lp); e.load_sampler("unipc", 20)
c = E.DevBuf.from_numpy(hash_normal("c", (B, 256, T))); p = E.DevBuf.from_numpy(hash_normal("p", (B, Lp, 256)))
m = E.DevBuf.from_numpy(np.ones((B, Lp), np.uint8))
x = E.DevBuf.from_numpy(hash_normal("x", (B, 100, T)))
s = E.Stream()
for _ in range(3): e.set_condition(c, p, m, stream=s)
s.sync()
e0, e1 = E.Event(), E.Event()
e0.record(s)
for _ in range(10): e.set_condition(c, p, m, stream=s)
e1.record(s); s.sync()
print("set_condition ms:", e0.elapsed_ms(e1) / 10)
e0.record(s); e.sample(x, use_graph=True, stream=s); e1.record(s); s.sync()
e0.record(s); e.sample(x, use_graph=True, stream=s); e1.record(s); s.sync()
print("20-step sample ms:", e0.elapsed_ms(e1))
