"""r5 session 3 diagnostics: is block b -> XCD b % 8 stable across launches; cooperative-prologue alone counts per plan."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import check
from ns2vc_amd import engine as E
from ns2vc_amd.spec import UNetConfig
from ns2vc_amd.weights import hash_normal, procedural_state_dict
lib = _lib.load()
def place(st, n=64, spin=0):
    out = (C.c_uint32 * (2 * n))()
    check(lib.ns2vc_debug_placement(st.ptr if st else None, n, spin, out), "placement")
    return np.frombuffer(out, dtype=np.uint32).reshape(n, 2)[:, 0].copy()
st = E.Stream()
for n in (8, 13, 64, 100, 7, 256, 9, 248, 64):
    x = place(st, n)
    print(f"launch of {n:4d} blocks: first 16 XCC ids {x[:16].tolist()}  ids == i%8: {bool((x == np.arange(n) % 8).all())}")
print("xcd_round_robin:", E.xcd_round_robin())
cfg = UNetConfig()
W = procedural_state_dict(cfg, 0)
B, T, Lp = 32, 938, 469
x = hash_normal("d.x", (B, cfg.latent_channels, T)); c = hash_normal("d.c", (B, cfg.content_channels, T)); p = hash_normal("d.p", (B, Lp, cfg.cross_attention_dim))
t = np.full((B,), 499.5, np.float32)
eng = E.Engine(cfg, precision="fp16"); eng.load_state_dict(W)
d_x, d_c, d_p, d_t, d_o = E.DevBuf.from_numpy(x), E.DevBuf.from_numpy(c), E.DevBuf.from_numpy(p), E.DevBuf.from_numpy(t), E.DevBuf(x.nbytes)
for ts in (1, 0):
    eng.set_option("conv_ts", bool(ts)); eng.prepare(B, T, Lp)
    eng.set_condition(d_c, d_p, None, stream=st)
    eng.gn_coop_alone(stream=st)
    for _ in range(3):
        eng.forward(d_x, d_t, d_o, stream=st)
    st.sync()
    print(f"conv_ts={ts}: prologue workgroups alone over 3 forwards: {eng.gn_coop_alone(stream=st)}; launches {eng.launches()}")
    ms = eng.profile_forward(reps=4, stream=st)
    ops = eng.op_info(0)
    for (nm, kind, fl, by), m in zip(ops, ms):
        if "down_blocks.3.resnets.1.conv1" in nm or "up_blocks.0.resnets.0.conv1" in nm or "down_blocks.0.resnets.0.conv1" in nm or "up_blocks.1.resnets.0.conv1" in nm:
            print(f"   {nm}: {m*1e3:.1f} us")
