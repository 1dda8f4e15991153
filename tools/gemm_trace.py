#!/usr/bin/env python3
"""Per-workgroup phase timing (s_memtime) of one GEMM shape: where do a block's microseconds go?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import GemmArgs, check
from ns2vc_amd.engine import DevBuf, sync

lib = _lib.load()
PREC = 2          # fp16 operands
cases = [  # (name, M, N, K, taps, geglu, output, (bm, bn, stages)); stages 12 / 13 = the 8-wave K-split kernel with ring 2 / 3
         ("L0.conv3", 30016, 128, 384, 3, 0, "f32", (64, 128, 13)), ("L0.conv3", 30016, 128, 384, 3, 0, "f32", (128, 128, 13)),
         ("L0.conv3cat", 30016, 128, 768, 3, 0, "f32", (64, 128, 13)), ("L0.conv3cat", 30016, 128, 768, 3, 0, "f32", (128, 128, 13)),
         ("L0.lin+res", 30016, 128, 128, 1, 0, "f32", (64, 128, 13)), ("L0.qkv", 30016, 384, 128, 1, 0, "op", (128, 128, 12)),
         ("L1.conv3", 15008, 256, 768, 3, 0, "f32", (128, 128, 13)), ("L1.conv3", 15008, 256, 768, 3, 0, "f32", (64, 128, 13)),
         ("L1.lin", 15008, 256, 256, 1, 0, "f32", (64, 128, 13)),
         ("L2.conv3", 7520, 384, 1152, 3, 0, "f32", (128, 128, 13)), ("L2.conv3", 7520, 384, 1152, 3, 0, "f32", (64, 128, 13)),
         ("L3.conv3big", 3776, 512, 3072, 3, 0, "f32", (64, 128, 13)),
         ("tiny.conv3", 192, 512, 1536, 3, 0, "f32", (64, 128, 13)), ("tiny.lin", 192, 128, 128, 1, 0, "f32", (64, 128, 13))]
for name, M, N, K, taps, geglu, outk, cfg in cases:
    Cin = K // taps; Tt = M // 32; Bb = 32; M = Bb * Tt
    A = DevBuf(M * Cin * 2 + 4096); W = DevBuf(N * K * 2); bias = DevBuf.from_numpy(np.zeros(N, np.float32))
    Nout = N // 2 if geglu else N
    O = DevBuf(M * Nout * 4)
    g = GemmArgs(); g.a0 = A.ptr; g.lda0 = Cin; g.c0 = Cin; g.B, g.Tin, g.Tout, g.M = Bb, Tt, Tt, M
    g.taps = taps; g.w = W.ptr; g.K = K; g.N = N; g.bias = bias.ptr; g.geglu = geglu
    if outk == "f32": g.out_f32 = O.ptr; g.ldo_f32 = Nout
    else: g.out_op = O.ptr; g.ldo_op = Nout
    nblk = (N // cfg[1]) * ((M + cfg[0] - 1) // cfg[0])
    T = DevBuf(nblk * 8 * 8)
    check(lib.ns2vc_debug_set_gemm_tile(*cfg), "tile")
    for _ in range(3): check(lib.ns2vc_k_gemm(C.byref(g), PREC, None), "gemm")
    sync()
    check(lib.ns2vc_debug_set_gemm_trace(T.ptr), "trace")
    check(lib.ns2vc_k_gemm(C.byref(g), PREC, None), "gemm"); sync()
    check(lib.ns2vc_debug_set_gemm_trace(None), "trace")
    t = T.to_numpy((nblk, 8), dtype=np.uint64).astype(np.float64)
    t0 = t[:, 0].min()
    d = np.diff(t[:, :7], axis=1)          # cycles of the 100 MHz? constant clock -> report raw + relative
    tot = t[:, 6] - t[:, 0]
    print(f"{name:12s} blocks={nblk:5d} cfg={cfg} kernel span={(t[:,6].max()-t0):9.0f} ticks; per-block total mean={tot.mean():8.0f} "
          f"| setup {d[:,0].mean():6.0f} issue {d[:,1].mean():6.0f} first-wait {d[:,2].mean():6.0f} kloop {d[:,3].mean():7.0f} "
          f"epi-stage {d[:,4].mean():6.0f} epi-store {d[:,5].mean():6.0f} | start spread {np.percentile(t[:,0]-t0,[50,90,100]).round(0)}")
lib.ns2vc_debug_set_gemm_tile(0, 0, 0)
