#!/usr/bin/env python3
"""The in-situ repro of round 3's non-deterministic GroupNorm prologue (profiles/r04_gn_prologue_rootcause.txt): the fused launch at the bench
shape, repeated, against ns2vc_k_groupnorm_stats + the same GEMM, bit for bit (GPU box).  22 of 22 launches failed before the r4 fix; the
instrumented variants it was run on (NS2VC_GNP_DETECT / NS2VC_GNP_FIX builds) are in the history at commit 1694573."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import GemmArgs, check
from ns2vc_amd.engine import DevBuf, sync

lib = _lib.load()
prec = 2
rng = np.random.default_rng(0)
for (B, T, Cc, N, taps) in ((32, 938, 128, 128, 3),):
    M, K = B * T, taps * Cc
    x = rng.standard_normal((B, T, Cc)).astype(np.float32)
    gam, bet = (1.0 + np.arange(Cc) / 256.0).astype(np.float32), (np.arange(Cc) / 64.0 + 0.25).astype(np.float32)
    blk = x.astype(np.float64).reshape(B, T, Cc // 16, 16)
    st = np.stack([np.rint(blk.sum(axis=(1, 3)) * 2.0 ** 28), np.rint((blk ** 2).sum(axis=(1, 3)) * 2.0 ** 16)], axis=-1).astype(np.int64)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    wp = C.c_void_p()
    check(lib.ns2vc_pack_weight(W.ctypes.data, N, K, prec, C.byref(wp)), "pack")
    d_x, d_g, d_b, d_st = DevBuf.from_numpy(x.reshape(M, Cc)), DevBuf.from_numpy(gam), DevBuf.from_numpy(bet), DevBuf.from_numpy(st)
    d_a0, d_a1 = DevBuf(M * Cc * 2), DevBuf(M * Cc * 2)
    d_o0, d_o1 = DevBuf(M * N * 4), DevBuf(M * N * 4)

    def args(a, o, fused):
        g = GemmArgs()
        g.a0 = a.ptr; g.lda0 = Cc; g.c0 = Cc
        g.B, g.Tin, g.Tout, g.M = B, T, T, M
        g.taps, g.tmode = taps, 0
        g.w = wp.value; g.K = K; g.N = N
        g.out_f32 = o.ptr; g.ldo_f32 = N
        if fused:
            g.gnp_x = d_x.ptr; g.gnp_ldx = Cc; g.gnp_stats = d_st.ptr; g.gnp_gamma = d_g.ptr; g.gnp_beta = d_b.ptr
            g.gnp_eps = 1e-5; g.gnp_G = 8; g.gnp_silu = 0
        return g
    g0, g1 = args(d_a0, d_o0, 0), args(d_a1, d_o1, 1)
    check(lib.ns2vc_k_groupnorm_stats(d_x.ptr, Cc, Cc, d_st.ptr, B, T, 8, 1e-5, d_g.ptr, d_b.ptr, None, 0, 0, 0, d_a0.ptr, prec, None), "gn")
    check(lib.ns2vc_k_gemm(C.byref(g0), prec, None), "gemm"); sync()
    ref, ref_a = d_o0.to_numpy((M, N)), d_a0.to_numpy((M, Cc), dtype=np.uint16)
    nrep = int(os.environ.get('GNP_REPS', '12'))
    nbad_total = 0
    for rep in range(nrep):
        d_a1.upload(np.full((M, Cc), 0x7e00 + rep, dtype=np.uint16))          # NaN pattern that changes every repetition
        d_o1.upload(np.zeros((M, N), np.float32))
        check(lib.ns2vc_k_gemm(C.byref(g1), prec, None), "gemm"); sync()
        out, a = d_o1.to_numpy((M, N)), d_a1.to_numpy((M, Cc), dtype=np.uint16)
        bad_rows = np.unique(np.argwhere(out != ref)[:, 0]) if not np.array_equal(out, ref, equal_nan=True) else np.array([], dtype=int)
        bad_a = np.unique(np.argwhere(a != ref_a)[:, 0])
        nbad_total += int(len(bad_a) > 0)
        if hasattr(lib, "ns2vc_debug_gnp_dump"):
            dbg = (C.c_uint32 * 248)()
            lib.ns2vc_debug_gnp_dump(dbg, 248)
            if dbg[0]:
                print(f"    detector: {dbg[0]} lanes whose gamma changed between the counted wait and the read after s_sleep")
                for sl in range(min(dbg[0], 6)):
                    r = dbg[8 + sl * 12: 8 + sl * 12 + 12]
                    print(f"      block {r[0]} tid {r[1]} (wave {r[1] >> 6} lane {r[1] & 63}) col {r[2]} m0 {r[3]}: early {[hex(v) for v in r[4:8]]} late {[hex(v) for v in r[8:12]]}")
        print(f"B={B} T={T} C={Cc} N={N} rep {rep}: result rows differing {len(bad_rows)} (first {bad_rows[:12].tolist()}, mod 64: {sorted(set((bad_rows % 64).tolist()))[:16]})  "
              f"operand rows differing {len(bad_a)} (first {bad_a[:12].tolist()})", flush=True)
        for r in bad_a[:3]:
            cols = np.argwhere(a[r] != ref_a[r])[:, 0]
            xr = x.reshape(M, Cc)[r, cols[:4]]
            print(f"    x there {xr.tolist()}  beta there {bet[cols[:4]].tolist()} gamma there {gam[cols[:4]].tolist()}")
            print(f"    row {r}: {len(cols)} of {Cc} columns differ (first {cols[:8].tolist()}); got {[hex(v) for v in a[r, cols[:6]]]} want {[hex(v) for v in ref_a[r, cols[:6]]]}  "
                  f"got as f16 {a[r, cols[:4]].view(np.float16).tolist()} want {ref_a[r, cols[:4]].view(np.float16).tolist()}", flush=True)
    print(f'launches with wrong operand rows: {nbad_total} of {nrep}')
    lib.ns2vc_dev_free(wp)
