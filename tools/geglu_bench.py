#!/usr/bin/env python3
"""Time the token-stationary GEGLU projection (csrc/geglu.hip) beside the GEMM it replaces, in isolation (GPU box only).

    python tools/geglu_bench.py [--prec fp16] [--reps 40] [--rotate 12]

M = 7520 rows (32 x 235 frames), dim 384.  --rotate N cycles over N copies of the activation rows and the result, so every launch reads rows that
are not in L2 (as inside the captured step).  With NS2VC_LIB pointing at a diagnostic build (make DEFS=-DNS2VC_GG_ABLATE=n) only the time means anything.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib                                # noqa: E402
from ns2vc_amd._lib import GegluArgs, GemmArgs, check    # noqa: E402
from ns2vc_amd.engine import DevBuf, Event, Stream       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--rotate", type=int, default=12)
    ap.add_argument("--rows", type=int, default=7520)
    a = ap.parse_args()
    prec = {"bf16": 1, "fp16": 2}[a.prec]
    lib = _lib.load()
    st = Stream()
    d, M = 384, a.rows
    rng = np.random.default_rng(0)
    W = (rng.standard_normal((8 * d, d)) / np.sqrt(d)).astype(np.float32)
    b = (0.3 * rng.standard_normal(8 * d)).astype(np.float32)
    stream, consts, wg = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(lib.ns2vc_pack_geglu(W.ctypes.data, b.ctypes.data, d, prec, C.byref(stream), C.byref(consts)), "pack_geglu")
    check(lib.ns2vc_pack_weight(W.ctypes.data, 8 * d, d, prec, C.byref(wg)), "pack_weight")
    y = rng.standard_normal((M, d)).astype(np.float32)
    ys = y.astype(np.float64).reshape(M, d // 64, 64)
    stats = DevBuf.from_numpy(np.stack([ys.sum(2), (ys ** 2).sum(2)], axis=-1).astype(np.float32))
    wsum = DevBuf.from_numpy(W.sum(1).astype(np.float32))
    bias = DevBuf.from_numpy(b)
    Ys, Hs = [], []
    for _ in range(a.rotate):
        p = C.c_void_p()
        check(lib.ns2vc_to_operand(y.ctypes.data, y.size, prec, C.byref(p)), "to_operand")
        Ys.append(p.value)
        Hs.append(DevBuf(M * 4 * d * 2))

    def timed(launch):
        for i in range(3):
            launch(i)
        e0, e1 = Event(), Event()
        e0.record(st)
        for i in range(a.reps):
            launch(i)
        e1.record(st)
        st.sync()
        return e0.elapsed_ms(e1) * 1e3 / a.reps

    f = GegluArgs()
    f.ldy = d; f.ln_stats = stats.ptr; f.ln_eps = 1e-5
    f.wstream = stream.value; f.consts = consts.value
    f.ldo = 4 * d; f.M = M; f.dim = d

    def k_geglu(i):
        f.yn = Ys[i % a.rotate]; f.out_op = Hs[i % a.rotate].ptr
        check(lib.ns2vc_k_geglu(C.byref(f), prec, st.ptr), "k_geglu")

    g = GemmArgs()
    g.lda0 = d; g.c0 = d
    g.B, g.Tin, g.Tout, g.M = 32, M // 32, M // 32, (M // 32) * 32
    g.taps, g.tmode = 1, 0
    g.w = wg.value; g.K = d; g.N = 8 * d; g.bias = bias.ptr
    g.geglu = 1; g.ldo_op = 4 * d
    g.ln_stats = stats.ptr; g.ln_wsum = wsum.ptr; g.ln_eps = 1e-5; g.ln_dim = d

    def k_gemm(i):
        g.a0 = Ys[i % a.rotate]; g.out_op = Hs[i % a.rotate].ptr
        check(lib.ns2vc_k_gemm(C.byref(g), prec, st.ptr), "k_gemm")

    fl = 2.0 * M * 8 * d * d
    t1 = timed(k_geglu)
    t0 = timed(k_gemm) if M % 32 == 0 else float("nan")
    t1b = timed(k_geglu)
    print(f"geglu M={M} dim={d} {a.prec} rotate={a.rotate}: token-stationary {t1:6.1f} us ({fl / t1 / 1e6:5.0f} TF/s) again {t1b:6.1f} us | GEMM (128x128 tiles, LayerNorm by linearity) {t0:6.1f} us ({fl / t0 / 1e6:5.0f} TF/s)"
          + (f"  [NS2VC_LIB={os.environ['NS2VC_LIB']}]" if os.environ.get("NS2VC_LIB") else ""))


if __name__ == "__main__":
    main()
