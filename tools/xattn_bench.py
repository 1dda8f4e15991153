#!/usr/bin/env python3
"""r6: the fused feed-forward kernel with the prompt cross-attention inside (ns2vc_ffn_args.att_*) beside the two launches it replaces (ns2vc_k_attention +
the pre-stage kernel), isolated, at the bench shapes of levels 0 and 1.  With NS2VC_LIB pointing at a -DNS2VC_XATT_ABLATE=n build: what the phase's parts cost."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import AttnArgs, FfnArgs, check
from ns2vc_amd.engine import DevBuf, Event, Stream, sync

lib = _lib.load()
prec = 2
reps = int(os.environ.get("REPS", "30"))
for (d, B, T, Lk) in ((128, 32, 938, 469), (256, 32, 469, 469)):
    rng = np.random.default_rng(d)
    M, hd = B * T, d // 8
    f16 = lambda a: np.ascontiguousarray(a.astype(np.float16))
    q, kv = DevBuf.from_numpy(f16(rng.standard_normal((M, d)))), DevBuf.from_numpy(f16(rng.standard_normal((B * Lk, 2 * d))))
    W1p, w2f, Wo = (rng.standard_normal((8 * d, d)) / np.sqrt(d)).astype(np.float32), (rng.standard_normal((d, 5 * d)) / np.sqrt(d)).astype(np.float32), (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
    stream = C.c_void_p()
    check(lib.ns2vc_pack_ffn_pre(W1p.ctypes.data, w2f.ctypes.data, Wo.ctypes.data, d, prec, C.byref(stream)), "pack")
    consts, bias2, x, bo, yp = (DevBuf.from_numpy(a.astype(np.float32)) for a in (np.ones((8 * d, 2)), np.zeros(d), rng.standard_normal((M, d)), np.zeros(d), rng.standard_normal((M, d))))
    img = DevBuf(int(lib.ns2vc_xattn_pack_bytes(B, Lk, hd)))
    check(lib.ns2vc_k_xattn_pack(kv.ptr, 2 * d, kv.ptr + 2 * d, 2 * d, B, Lk, hd, img.ptr, prec, None), "xattn_pack")
    out, ao = DevBuf(M * d * 4), DevBuf(M * d * 2)

    def ffn(att):
        f = FfnArgs()
        f.wstream = stream.value; f.consts = consts.ptr; f.bias2 = bias2.ptr; f.res = x.ptr; f.ldres = d
        f.out_f32 = out.ptr; f.ldo_f32 = d; f.B, f.T, f.M, f.dim = B, T, M, d; f.ln_eps = 1e-5
        f.pre_bias = bo.ptr; f.pre_res = yp.ptr; f.pre_ldres = d
        if att:
            f.att_q = q.ptr; f.att_ldq = d; f.att_kv = img.ptr; f.att_scale = 1.0 / np.sqrt(hd); f.att_Lk = Lk
        else:
            f.pre_a = ao.ptr; f.pre_lda = d
        return f
    a = AttnArgs()
    a.q = q.ptr; a.k = kv.ptr; a.v = kv.ptr + 2 * d; a.ldq = d; a.ldk = 2 * d; a.ldv = 2 * d
    a.B, a.H, a.Lq, a.Lk = B, 8, T, Lk; a.scale = 1.0 / np.sqrt(hd); a.out = ao.ptr; a.ldo = d
    st = Stream()
    res = {}
    for name, fn in (("attention alone", lambda: check(lib.ns2vc_k_attention(C.byref(a), hd, prec, st.ptr), "attn")),
                     ("ffn (pre-stage) alone", lambda f=ffn(0): check(lib.ns2vc_k_ffn(C.byref(f), prec, st.ptr), "ffn")),
                     ("ffn with the attention inside", lambda f=ffn(1): check(lib.ns2vc_k_ffn(C.byref(f), prec, st.ptr), "ffn+att"))):
        for _ in range(3):
            fn()
        e0, e1 = Event(), Event()
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        st.sync()
        res[name] = e0.elapsed_ms(e1) * 1e3 / reps
    print(f"dim {d} B {B} T {T} Lk {Lk}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in res.items()) +
          f"  | separate {res['attention alone'] + res['ffn (pre-stage) alone']:6.1f} vs fused {res['ffn with the attention inside']:6.1f}", flush=True)
    lib.ns2vc_dev_free(stream)
