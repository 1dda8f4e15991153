#!/usr/bin/env python3
"""Per-workgroup phase timing of the row-chain kernel (GPU box; needs a `make TRACE=1` build in NS2VC_LIB).
    make -C ns2vc_amd/csrc TRACE=1 OUT=../lib/variants/trace -j8
    NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/trace/libns2vc_hip.so python tools/rowchain_trace.py
Shapes: the four chains of the 10 s x batch-32 plan (dim 128 / 256, q|k|v and to_q), plain A operand and GroupNorm prologue.
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import RowchainArgs, check
from ns2vc_amd.engine import DevBuf, Event, Stream

lib = _lib.load()
rng = np.random.default_rng(0)
prec = 2
dims = [int(v) for v in os.environ.get("RC_DIMS", "128,256").split(",")]
for d in dims:
    B, T = 32, {128: 938, 256: 469, 384: 235, 512: 118}[d]
    M = B * T
    for mult, gn in ((3, True), (1, False)):
        n2 = mult * d
        W1 = (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
        W2 = (rng.standard_normal((n2, d)) / np.sqrt(d)).astype(np.float32)
        stream = C.c_void_p()
        check(lib.ns2vc_pack_rowchain(W1.ctypes.data, W2.ctypes.data, d, n2, prec, C.byref(stream)), "pack")
        x = rng.standard_normal((M, d)).astype(np.float32)
        p = C.c_void_p()
        check(lib.ns2vc_to_operand(x.ctypes.data, x.size, prec, C.byref(p)), "to_op")
        d_x = DevBuf.from_numpy(x)
        blk = x.astype(np.float64).reshape(B, T, d // 16, 16)
        st = DevBuf.from_numpy(np.stack([np.rint(blk.sum(axis=(1, 3)) * 2.0 ** 28), np.rint((blk ** 2).sum(axis=(1, 3)) * 2.0 ** 16)], axis=-1).astype(np.int64))
        gam, bet = DevBuf.from_numpy(np.ones(d, np.float32)), DevBuf.from_numpy(np.zeros(d, np.float32))
        b1 = DevBuf.from_numpy(np.zeros(d, np.float32))
        consts = DevBuf.from_numpy(np.zeros((n2, 2), np.float32))
        res = DevBuf.from_numpy(rng.standard_normal((M, d)).astype(np.float32))
        y = DevBuf(M * d * 4)
        z = DevBuf(M * n2 * 2)
        f = RowchainArgs()
        f.a_op = None if gn else p.value; f.lda = d; f.wstream = stream.value; f.bias1 = b1.ptr; f.consts2 = consts.ptr
        f.res = None if (gn or os.environ.get('RC_NO_RES')) else res.ptr; f.ldres = d; f.out1_f32 = None if os.environ.get("RC_NO_Y") else y.ptr; f.ldo1 = d; f.out2_op = z.ptr; f.ldo2 = n2
        f.ln_eps = 1e-5; f.M = M; f.dim = d; f.n2 = n2; f.ln_health = None
        if gn:
            f.gn_x = d_x.ptr; f.ldx = d; f.gn_stats = st.ptr; f.gn_gamma = gam.ptr; f.gn_beta = bet.ptr; f.gn_eps = 1e-6; f.T = T; f.G = 8
        s = Stream()
        for _ in range(3):
            check(lib.ns2vc_k_rowchain(C.byref(f), prec, s.ptr), "rowchain")
        e0, e1 = Event(), Event()
        e0.record(s)
        for _ in range(20):
            check(lib.ns2vc_k_rowchain(C.byref(f), prec, s.ptr), "rowchain")
        e1.record(s); s.sync()
        us = e0.elapsed_ms(e1) * 1e3 / 20
        nblk = (M + 63) // 64
        trace = DevBuf.from_numpy(np.zeros((nblk, 8), np.uint64))
        check(lib.ns2vc_debug_set_gemm_trace(trace.ptr), "trace")
        check(lib.ns2vc_k_rowchain(C.byref(f), prec, s.ptr), "rowchain"); s.sync()
        lib.ns2vc_debug_set_gemm_trace(None)
        t = trace.to_numpy((nblk, 8), dtype=np.uint64).astype(np.float64)
        t = t[t[:, 5] > 0]                      # (128-token workgroups: half the blocks)
        t0 = t[:, 0].min()
        ph = [np.median(t[:, i + 1] - t[:, i]) for i in range(5)]
        print(f"dim {d} n2 {n2} {'GroupNorm prologue' if gn else 'operand A + residual'} M {M}: {len(t)} workgroups, launch {us:.1f} us; per-workgroup cycles (median): "
              f"prologue {ph[0]:.0f}  stage-1 loop {ph[1]:.0f}  stage-1 epilogue {ph[2]:.0f}  stage-2 loop {ph[3]:.0f}  stage-2 epilogue + drain {ph[4]:.0f}  "
              f"total {np.median(t[:, 5] - t[:, 0]):.0f}; first start..last end {(t[:, 5].max() - t0):.0f}; start spread {(t[:, 0].max() - t0):.0f}")
