#!/usr/bin/env python3
"""r4: per-workgroup phase stamps (s_memtime, `make TRACE=1` build) of the conv GEMMs WITH and WITHOUT the GroupNorm prologue, the
resnet shapes of the bench workload: what the prologue costs inside the launch (the `issue` phase holds it) and per launch; `+coop` = with ns2vc_gemm_args.gnp_sync
(the column tiles of a row block share the rows).

    make -C ns2vc_amd/csrc TRACE=1 OUT=../lib/variants/trace && NS2VC_LIB=ns2vc_amd/lib/variants/trace/libns2vc_hip.so python tools/gnp_trace.py
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import GemmArgs, check
from ns2vc_amd.engine import DevBuf, sync, Event

lib = _lib.load()
PREC = 2
cases = [  # (name, frames per item, C in, N out, temb)
    ("L0 conv 128->128", 938, 128, 128), ("L1 conv 256->256", 469, 256, 256), ("L2 conv 384->384", 235, 384, 384), ("L3 conv 512->512", 118, 512, 512)]
B = 32
for name, T, Cc, N in cases:
    M, K = B * T, 3 * Cc
    A = DevBuf(M * Cc * 2 + 4096); W = DevBuf(N * K * 2); bias = DevBuf.from_numpy(np.zeros(N, np.float32))
    X = DevBuf.from_numpy(np.random.default_rng(0).standard_normal((M, Cc)).astype(np.float32))
    St = DevBuf.from_numpy(np.zeros((B, Cc // 16, 2), np.int64)); Ga = DevBuf.from_numpy(np.ones(Cc, np.float32)); Be = DevBuf.from_numpy(np.zeros(Cc, np.float32))
    O = DevBuf(M * N * 4)
    Sy = DevBuf.from_numpy(np.zeros((M + 63) // 64, np.uint64))
    for fused in (0, 1, 2):
        g = GemmArgs(); g.a0 = A.ptr; g.lda0 = Cc; g.c0 = Cc; g.B, g.Tin, g.Tout, g.M = B, T, T, M
        g.taps = 3; g.w = W.ptr; g.K = K; g.N = N; g.bias = bias.ptr; g.out_f32 = O.ptr; g.ldo_f32 = N
        if fused:
            g.gnp_sync = Sy.ptr if fused == 2 else None
            g.gnp_x = X.ptr; g.gnp_ldx = Cc; g.gnp_stats = St.ptr; g.gnp_gamma = Ga.ptr; g.gnp_beta = Be.ptr; g.gnp_eps = 1e-5; g.gnp_G = 8; g.gnp_silu = 1
        for _ in range(3): check(lib.ns2vc_k_gemm(C.byref(g), PREC, None), "gemm")
        sync()
        e0, e1 = Event(), Event()
        e0.record(None)
        for _ in range(20): check(lib.ns2vc_k_gemm(C.byref(g), PREC, None), "gemm")
        e1.record(None); sync()
        us = e0.elapsed_ms(e1) * 1000 / 20
        nblk = 4096
        Tr = DevBuf(nblk * 8 * 8); Tr.upload(np.zeros((nblk, 8), np.uint64))
        check(lib.ns2vc_debug_set_gemm_trace(Tr.ptr), "trace")
        check(lib.ns2vc_k_gemm(C.byref(g), PREC, None), "gemm"); sync()
        check(lib.ns2vc_debug_set_gemm_trace(None), "trace")
        t = Tr.to_numpy((nblk, 8), dtype=np.uint64).astype(np.float64)
        t = t[t[:, 6] > 0]                          # (workgroups that padded a cooperative grid leave at once)
        label = ('plain', '+norm', '+coop')[fused]
        t0 = t[:, 0].min()
        d = np.diff(t[:, :7], axis=1)               # s_memtime ticks (shader clock, ~2.4 per ns)
        print(f"{name:18s} {label:5s} {us:6.1f} us/launch back to back | blocks {len(t):4d} span {(t[:, 6].max() - t0):7.0f} clk | per block (clk): "
              f"setup {d[:, 0].mean():5.0f} issue(+prologue) {d[:, 1].mean():5.0f} first-wait {d[:, 2].mean():5.0f} kloop {d[:, 3].mean():6.0f} "
              f"epi-stage {d[:, 4].mean():5.0f} epi-store {d[:, 5].mean():5.0f}")
