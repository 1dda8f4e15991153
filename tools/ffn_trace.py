#!/usr/bin/env python3
"""Per-workgroup phase timing of the fused feed-forward kernel (GPU box; needs a `make TRACE=1` build in NS2VC_LIB).
    make -C ns2vc_amd/csrc TRACE=1 OUT=../lib/variants/trace -j8
    NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/trace/libns2vc_hip.so python tools/ffn_trace.py
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import FfnArgs, check
from ns2vc_amd.engine import DevBuf, Event, Stream

lib = _lib.load()
rng = np.random.default_rng(0)
prec = 2
for d, B, T in ((128, 32, 938), (256, 32, 469)):
    M = B * T
    W1p = (rng.standard_normal((8 * d, d)) / np.sqrt(d)).astype(np.float32)
    w2f = (rng.standard_normal((d, 5 * d)) / np.sqrt(5 * d)).astype(np.float32)
    stream = C.c_void_p()
    check(lib.ns2vc_pack_ffn(W1p.ctypes.data, w2f.ctypes.data, d, prec, C.byref(stream)), "pack")
    y = rng.standard_normal((M, d)).astype(np.float32)
    p = C.c_void_p()
    check(lib.ns2vc_to_operand(y.ctypes.data, y.size, prec, C.byref(p)), "to_op")
    ys = y.astype(np.float64).reshape(M, d // 64, 64)
    stats = DevBuf.from_numpy(np.stack([ys.sum(2), (ys ** 2).sum(2)], -1).astype(np.float32))
    consts = DevBuf.from_numpy(rng.standard_normal((8 * d, 2)).astype(np.float32) * 0.1)
    bias2 = DevBuf.from_numpy(np.zeros(d, np.float32))
    x = DevBuf.from_numpy(rng.standard_normal((M, d)).astype(np.float32))
    out = DevBuf(M * d * 4)
    gs = DevBuf.from_numpy(np.zeros((B, d // 16, 2), np.int64))
    nblk = (M + 63) // 64
    trace = DevBuf.from_numpy(np.zeros((nblk, 8), np.uint64))
    f = FfnArgs()
    f.yn = p.value; f.ldy = d; f.ln_stats = stats.ptr; f.ln_eps = 1e-5
    f.wstream = stream.value; f.consts = consts.ptr; f.bias2 = bias2.ptr
    f.res = x.ptr; f.ldres = d; f.out_f32 = out.ptr; f.ldo_f32 = d; f.stats = gs.ptr
    f.B, f.T, f.M, f.dim = B, T, M, d
    st = Stream()
    for _ in range(3):
        check(lib.ns2vc_k_ffn(C.byref(f), prec, st.ptr), "ffn")
    e0, e1 = Event(), Event()
    e0.record(st)
    for _ in range(20):
        check(lib.ns2vc_k_ffn(C.byref(f), prec, st.ptr), "ffn")
    e1.record(st); st.sync()
    us = e0.elapsed_ms(e1) * 1e3 / 20
    check(lib.ns2vc_debug_set_gemm_trace(trace.ptr), "trace")
    check(lib.ns2vc_k_ffn(C.byref(f), prec, st.ptr), "ffn"); st.sync()
    lib.ns2vc_debug_set_gemm_trace(None)
    t = trace.to_numpy((nblk, 8), dtype=np.uint64).astype(np.float64)
    t0 = t[:, 0].min()
    pro, ff1, gg, ff2, po, epi = t[:, 1] - t[:, 0], t[:, 2], t[:, 3], t[:, 4], t[:, 5], t[:, 6]
    tot = t[:, 7] - t[:, 0]
    print(f"dim {d} M {M}: {nblk} workgroups, kernel {us:.1f} us; per-workgroup cycles (median): prologue {np.median(pro):.0f}  ff.net.0 {np.median(ff1):.0f}  "
          f"GEGLU {np.median(gg):.0f}  W2' {np.median(ff2):.0f}  Wpo {np.median(po):.0f}  epilogue {np.median(epi):.0f}  total {np.median(tot):.0f}; "
          f"first start..last end {(t[:, 7].max() - t0):.0f} cycles; start spread {(t[:, 0].max() - t0):.0f}")
