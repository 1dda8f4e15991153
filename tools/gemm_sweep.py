#!/usr/bin/env python3
"""Time the model's GEMM shapes under every tile / ring-depth configuration (GPU box only).

    python tools/gemm_sweep.py [--prec bf16] [--reps 30] > gpurun_out/gemm_sweep.txt

Drives ns2vc_k_gemm through the C ABI with HIP events; used to derive the tile heuristic
in csrc/gemm.hip.  Shapes: the distinct (M, N, K, taps, geglu, res, out) of the 10 s x batch-32 plan.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib                     # noqa: E402
from ns2vc_amd._lib import GemmArgs, check     # noqa: E402
from ns2vc_amd.engine import DevBuf, Event, Stream  # noqa: E402

# (BM, BN, stages | flags << 8); stages 2..4 = LDS-DMA ring depth of the 4-wave kernel (gemm2_kernel), 12 / 13 = the 8-wave
# K-split kernel (gemm4_kernel) with ring 2 / 3; flags (gemm2 only): 1 = K rotation, 2 = loads only, 4 = no steady-state loads
CONFIGS = [(128, 128, 12), (128, 128, 13), (64, 128, 12), (64, 128, 13), (64, 128, 2), (64, 64, 2), (64, 64, 3), (64, 64, 4)]
ABLATE = [(64, 128, 2), (64, 128, 2 | 256), (64, 128, 2 | 512), (64, 128, 2 | 1024), (64, 64, 2 | 256), (64, 64, 2 | 512), (64, 64, 2 | 1024)]


# gemm4_kernel ablations (variant library built with DEFS=-DNS2VC_GEMM_ABLATE=1): flags 2 = DMA only, 4 = no steady-state DMA,
# 8 = no fragment reads, 16 = no MFMAs; 4|8 = MFMAs only, 4|16 = fragment reads only
ABLATE4 = [(bm, 128, 13 | (f << 8)) for bm in (64, 128) for f in (0, 2, 4, 8, 16, 4 | 8, 4 | 16)]


# loader / consumer specialised gemm4 (stages 23 / 24 = ring 3 / 4) beside the plain K-split kernel
# (the 8 + 8 / 8 + 4 wave modes and ring 2 / 4 of profiles/r03_gemm_spec.txt need their NS2VC_CASE4S / NS2VC_SET4S lines back in gemm.hip)
SPEC = [(64, 128, 13), (64, 128, 23), (128, 128, 12), (128, 128, 13), (128, 128, 23)]
# r5: the tap-sharing conv kernel (convts.hip; stages 54 / 58 = 4 / 8 loader waves, BN = 64 / 128) beside gemm4's loader / consumer tiles, k = 3 shapes only
TS = [(64, 128, 23), (128, 128, 23), (128, 64, 54), (128, 128, 54), (128, 64, 58), (128, 128, 58), (128, 64, 64), (128, 64, 68)]   # 64 / 68: K-split consumers


def stride_shapes():
    """power-of-two vs odd-multiple row strides (L2 channel spread of the weight / activation rows)"""
    out = []
    for K in (512, 576, 1024, 1088, 1536, 1600, 2048, 2112, 3072, 3136):
        out.append((f"M3776 N512 lin K={K}", 32 * 118, 512, K, 1, 0, 0, "f32"))
    for K in (1152, 1216):
        out.append((f"M7520 N384 lin K={K}", 32 * 235, 384, K, 1, 0, 0, "f32"))
    return out


def shapes(B=32, T=938):
    Ts = [T, (T + 1) // 2, ((T + 1) // 2 + 1) // 2, (((T + 1) // 2 + 1) // 2 + 1) // 2]
    Cs = [128, 256, 384, 512]
    out = []
    for l, (Tl, c) in enumerate(zip(Ts, Cs)):
        M = B * Tl
        out += [(f"L{l}.conv3 {c}->{c}", M, c, 3 * c, 3, 0, 1, "f32"),
                (f"L{l}.lin {c}->{c} +res", M, c, c, 1, 0, 1, "f32"),
                (f"L{l}.lin {c}->{c} op", M, c, c, 1, 0, 0, "op"),
                (f"L{l}.qkv", M, 3 * c, c, 1, 0, 0, "op"),
                (f"L{l}.geglu", M, 8 * c, c, 1, 1, 0, "op"),
                (f"L{l}.ff_out", M, c, 4 * c, 1, 0, 1, "op")]
    out += [("L3.conv3 1024->512", B * Ts[3], 512, 3 * 1024, 3, 0, 0, "f32"), ("L2.conv3 896->384", B * Ts[2], 384, 3 * 896, 3, 0, 0, "f32"),
            ("L1.conv3 640->256", B * Ts[1], 256, 3 * 640, 3, 0, 0, "f32"), ("L0.conv3 384->128", B * Ts[0], 128, 3 * 384, 3, 0, 0, "f32"),
            ("L3.sc 1024->512", B * Ts[3], 512, 1024, 1, 0, 0, "f32"), ("temb", B, 14848, 512, 1, 0, 0, "f32")]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--ablate", action="store_true", help="time the ABLATE list (K rotation / loads-only / compute-only variants)")
    ap.add_argument("--ablate4", action="store_true", help="time the ABLATE4 list (gemm4_kernel with parts of its K loop removed; needs the NS2VC_GEMM_ABLATE build)")
    ap.add_argument("--rotate", type=int, default=1, help="cycle over this many copies of the operands (A, W): with enough copies to exceed the "
                                                        "32 MB of L2 every launch reads L2-cold data, as inside the captured step")
    ap.add_argument("--strides", action="store_true", help="time stride_shapes() instead of the plan's shapes")
    ap.add_argument("--spec", action="store_true", help="time the SPEC list (loader / consumer wave specialisation)")
    ap.add_argument("--ts", action="store_true", help="time the TS list (tap-sharing conv kernel) on the k = 3 shapes")
    a = ap.parse_args()
    global CONFIGS
    if a.ts:
        CONFIGS = TS
    if a.spec:
        CONFIGS = SPEC
    if a.ablate:
        CONFIGS = ABLATE
    if a.ablate4:
        CONFIGS = ABLATE4
    prec = {"fp32": 0, "bf16": 1, "fp16": 2}[a.prec]
    esz = 4 if prec == 0 else 2
    lib = _lib.load()
    st = Stream()
    rng = np.random.default_rng(0)
    print(f"# prec={a.prec}; time in us (best config marked *)")
    for name, M, N, K, taps, geglu, res, outk in (stride_shapes() if a.strides else shapes()):
        if a.ts and taps != 3:
            continue
        Cin = K // taps
        Tt = M // 32 if M >= 32 * 8 else 1
        Bb = M // Tt
        M = Bb * Tt
        As = [DevBuf(M * Cin * esz + 4096) for _ in range(a.rotate)]
        Ws = [DevBuf(N * K * esz) for _ in range(a.rotate)]
        A, W = As[0], Ws[0]
        bias = DevBuf.from_numpy(rng.standard_normal(N).astype(np.float32))
        Nout = N // 2 if geglu else N
        R = DevBuf(M * Nout * 4) if res else None
        O32 = DevBuf(M * Nout * 4)
        Oop = DevBuf(M * Nout * esz)
        g = GemmArgs()
        g.a0 = A.ptr; g.lda0 = Cin; g.c0 = Cin
        g.B, g.Tin, g.Tout, g.M = Bb, Tt, Tt, M
        g.taps, g.tmode = taps, 0
        g.w = W.ptr; g.K = K; g.N = N; g.bias = bias.ptr
        if R is not None:
            g.res = R.ptr; g.ldres = Nout
        g.geglu = geglu
        if outk == "f32":
            g.out_f32 = O32.ptr; g.ldo_f32 = Nout
        else:
            g.out_op = Oop.ptr; g.ldo_op = Nout
        flops = 2.0 * M * N * K
        row = []
        for cfg in CONFIGS:
            if N % cfg[1] or (geglu and cfg[1] != 128):
                row.append(None)
                continue
            check(lib.ns2vc_debug_set_gemm_tile(*cfg), "tile")
            for _ in range(3):
                check(lib.ns2vc_k_gemm(C.byref(g), prec, st.ptr), "gemm")
            e0, e1 = Event(), Event()
            e0.record(st)
            for i in range(a.reps):
                if a.rotate > 1:
                    g.a0 = As[i % a.rotate].ptr; g.w = Ws[i % a.rotate].ptr
                check(lib.ns2vc_k_gemm(C.byref(g), prec, st.ptr), "gemm")
            e1.record(st)
            st.sync()
            row.append(e0.elapsed_ms(e1) * 1e3 / a.reps)
        lib.ns2vc_debug_set_gemm_tile(0, 0, 0)
        for _ in range(3):
            check(lib.ns2vc_k_gemm(C.byref(g), prec, st.ptr), "gemm")
        e0, e1 = Event(), Event()
        e0.record(st)
        for i in range(a.reps):
            if a.rotate > 1:
                g.a0 = As[i % a.rotate].ptr; g.w = Ws[i % a.rotate].ptr
            check(lib.ns2vc_k_gemm(C.byref(g), prec, st.ptr), "gemm")
        e1.record(st)
        st.sync()
        heur = e0.elapsed_ms(e1) * 1e3 / a.reps
        best = min(v for v in row if v is not None)
        cells = " ".join(("   -- " if v is None else f"{v:6.1f}" + ("*" if v == best else " ")) for v in row)
        print(f"{name:22s} M={M:6d} N={N:5d} K={K:5d} | {cells} | heur {heur:6.1f} best {flops/best/1e6:6.0f} TF/s")
    print("# configs:", " ".join(f"{c[0]}x{c[1]}s{c[2] & 255}" + (f"f{c[2] >> 8}" if c[2] >> 8 else "") for c in CONFIGS))


if __name__ == "__main__":
    main()
