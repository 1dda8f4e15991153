#!/usr/bin/env python3
"""Per-workgroup phase timing (s_memtime) of the tap-sharing conv kernel beside gemm4's loader / consumer tile: where do a block's cycles go?
Needs a trace build (make TRACE=1 OUT=../lib/variants/trace) via NS2VC_LIB.  r5."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import GemmArgs, check
from ns2vc_amd.engine import DevBuf, sync

lib = _lib.load()
PREC = 2
shapes = [("L3.conv3 512", 118, 512, 512), ("L3.conv3 1024", 118, 1024, 512), ("L2.conv3 384", 235, 384, 384), ("L1.conv3 256", 469, 256, 256), ("L0.conv3 128", 938, 128, 128)]
rot = 8
GNP = "--gnp" in sys.argv          # with the cooperative GroupNorm prologue in front (zero statistics: rows become beta; timing only)
for name, T, Cin, N in shapes:
    B = 32; M = B * T; K = 3 * Cin
    As = [DevBuf(M * Cin * 2 + 4096) for _ in range(rot)]; Ws = [DevBuf(N * K * 2) for _ in range(rot)]
    for a in As: a.upload(np.zeros(M * Cin // 2 + 1024, np.float32))
    for w in Ws: w.upload(np.zeros(N * K // 2, np.float32))
    bias = DevBuf.from_numpy(np.zeros(N, np.float32)); O = DevBuf(M * N * 4)
    g = GemmArgs(); g.lda0 = Cin; g.c0 = Cin; g.B, g.Tin, g.Tout, g.M = B, T, T, M
    g.taps = 3; g.K = K; g.N = N; g.bias = bias.ptr; g.out_f32 = O.ptr; g.ldo_f32 = N
    if GNP:
        X = DevBuf(M * Cin * 4); X.upload(np.zeros(M * Cin, np.float32))
        St = DevBuf.from_numpy(np.zeros((B, Cin // 16, 2), dtype=np.int64))
        Ga, Be = DevBuf.from_numpy(np.ones(Cin, np.float32)), DevBuf.from_numpy(np.zeros(Cin, np.float32))
        Sy = DevBuf.from_numpy(np.zeros((M + 63) // 64, dtype=np.uint64))
        g.gnp_x = X.ptr; g.gnp_ldx = Cin; g.gnp_stats = St.ptr; g.gnp_gamma = Ga.ptr; g.gnp_beta = Be.ptr; g.gnp_eps = 1e-5; g.gnp_G = 8; g.gnp_silu = 1
        g.gnp_sync = Sy.ptr
    for cfg in [(64, 128, 23), (128, 64, 54), (128, 128, 54), (128, 64, 58)]:
        if N % cfg[1]:
            continue
        ts = cfg[2] >= 50
        nbm = (B * (T + 1) + 125) // 126 if ts else (M + cfg[0] - 1) // cfg[0]
        nblk = nbm * (N // cfg[1])
        if GNP and N // cfg[1] > 1:
            nblk = 8 * ((nbm + 7) // 8) * (N // cfg[1])          # the cooperative grid is padded to whole row blocks per XCD
        W = 16 if ts else 8
        Tr = DevBuf(nblk * W * 8)
        check(lib.ns2vc_debug_set_gemm_tile(*cfg), "tile")
        for i in range(rot + 2):
            g.a0 = As[i % rot].ptr; g.w = Ws[i % rot].ptr
            if GNP:
                Sy.upload(np.zeros((M + 63) // 64, dtype=np.uint64))
            check(lib.ns2vc_k_gemm(C.byref(g), PREC, None), "gemm")
        sync()
        Tr.upload(np.zeros(nblk * W, np.uint64))
        check(lib.ns2vc_debug_set_gemm_trace(Tr.ptr), "trace")
        g.a0 = As[2 % rot].ptr; g.w = Ws[2 % rot].ptr
        if GNP:
            Sy.upload(np.zeros((M + 63) // 64, dtype=np.uint64))
        check(lib.ns2vc_k_gemm(C.byref(g), PREC, None), "gemm"); sync()
        check(lib.ns2vc_debug_set_gemm_trace(None), "trace")
        t = Tr.to_numpy((nblk, W), dtype=np.uint64).astype(np.float64)
        t = t[t[:, (3 if ts else 6)] > 0]                         # (padding workgroups of a cooperative grid return at once)
        if len(t) == 0:                                           # (this object was not built with TRACE=1)
            continue
        t0 = t[:, 0].min()
        if ts:
            steps = 3 * (Cin // 64)
            loop = t[:, 2] - t[:, 1]
            print(f"{name:14s} cfg={cfg} blocks={len(t):4d} span={(t[:,3].max()-t0):8.0f} | setup+prologue {(t[:,7]-t[:,0]).mean():6.0f} first issue {(t[:,1]-t[:,7]).mean():5.0f} loop {loop.mean():7.0f} ({loop.mean()/steps:5.0f}/step x {steps}) "
                  f"epilogue {(t[:,3]-t[:,2]).mean():6.0f} | loader: counted waits {t[:,4].mean():7.0f} barrier {t[:,5].mean():7.0f} issue {t[:,6].mean():7.0f} | "
                  f"consumer: barrier {t[:,8].mean():7.0f} reads+mfma {t[:,9].mean():7.0f}")
        else:
            d = np.diff(t[:, :7], axis=1)
            tiles = K // 64
            print(f"{name:14s} cfg={cfg} blocks={len(t):4d} span={(t[:,6].max()-t0):8.0f} | setup {d[:,0].mean():6.0f} (prologue +) issue {d[:,1].mean():6.0f} first-wait {d[:,2].mean():6.0f} "
                  f"kloop {d[:,3].mean():7.0f} ({d[:,3].mean()/tiles:5.0f}/tile x {tiles}) epi-stage {d[:,4].mean():6.0f} epi-store {d[:,5].mean():6.0f}")
    lib.ns2vc_debug_set_gemm_tile(0, 0, 0)
