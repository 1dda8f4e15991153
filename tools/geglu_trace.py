#!/usr/bin/env python3
"""Per-workgroup phase timing (cycle counter, wave 0) of the token-stationary GEGLU kernel: where do a block's cycles go?
Needs a trace build (make TRACE=1 OUT=../lib/variants/trace) via NS2VC_LIB.  r5."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import GegluArgs, check
from ns2vc_amd.engine import DevBuf, sync

lib = _lib.load()
PREC, d, M, rot = 2, 384, 7520, 12
rng = np.random.default_rng(0)
W = (rng.standard_normal((8 * d, d)) / np.sqrt(d)).astype(np.float32)
b = (0.3 * rng.standard_normal(8 * d)).astype(np.float32)
stream, consts = C.c_void_p(), C.c_void_p()
check(lib.ns2vc_pack_geglu(W.ctypes.data, b.ctypes.data, d, PREC, C.byref(stream), C.byref(consts)), "pack_geglu")
y = rng.standard_normal((M, d)).astype(np.float32)
ys = y.astype(np.float64).reshape(M, d // 64, 64)
stats = DevBuf.from_numpy(np.stack([ys.sum(2), (ys ** 2).sum(2)], axis=-1).astype(np.float32))
Ys, Hs = [], []
for _ in range(rot):
    p = C.c_void_p()
    check(lib.ns2vc_to_operand(y.ctypes.data, y.size, PREC, C.byref(p)), "to_operand")
    Ys.append(p.value); Hs.append(DevBuf(M * 4 * d * 2))
f = GegluArgs()
f.ldy = d; f.ln_stats = stats.ptr; f.ln_eps = 1e-5; f.wstream = stream.value; f.consts = consts.value; f.ldo = 4 * d; f.M = M; f.dim = d
nblk = ((M + 127) // 128) * 4
Tr = DevBuf(nblk * 16 * 8)
for i in range(rot + 2):
    f.yn = Ys[i % rot]; f.out_op = Hs[i % rot].ptr
    check(lib.ns2vc_k_geglu(C.byref(f), PREC, None), "k_geglu")
sync()
Tr.upload(np.zeros(nblk * 16, np.uint64))
check(lib.ns2vc_debug_set_gemm_trace(Tr.ptr), "trace")
f.yn = Ys[2]; f.out_op = Hs[2].ptr
check(lib.ns2vc_k_geglu(C.byref(f), PREC, None), "k_geglu"); sync()
check(lib.ns2vc_debug_set_gemm_trace(None), "trace")
t = Tr.to_numpy((nblk, 16), dtype=np.uint64).astype(np.float64)
t0 = t[:, 0].min()
tot = t[:, 7] - t[:, 0]
names = ["prologue (token rows, first ring, statistics)", "wait for the tile (vmcnt)", "barrier (+ own LDS reads done)", "tile issue + fragment reads issued",
         "MFMAs + GEGLU chunk, to completion", "stores"]
cols = [t[:, 1] - t[:, 0]] + [t[:, 2 + i] for i in range(5)]
print(f"# geglu_kernel M={M}: {nblk} workgroups; entry spread {np.ptp(t[:, 0]):.0f}, exit - first entry max {np.max(t[:, 7]) - t0:.0f}, per-block total median {np.median(tot):.0f} (counter ticks)")
for n, c in zip(names, cols):
    print(f"  {n:48s} median {np.median(c):8.0f}  ({100 * np.median(c) / np.median(tot):5.1f} %)   min {c.min():8.0f} max {c.max():8.0f}")
print(f"  {'unaccounted':48s} median {np.median(tot - sum(cols)):8.0f}")
