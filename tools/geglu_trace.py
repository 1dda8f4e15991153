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
Tr = DevBuf(nblk * 8 * 16 * 8)
for i in range(rot + 2):
    f.yn = Ys[i % rot]; f.out_op = Hs[i % rot].ptr
    check(lib.ns2vc_k_geglu(C.byref(f), PREC, None), "k_geglu")
sync()
Tr.upload(np.zeros(nblk * 8 * 16, np.uint64))
check(lib.ns2vc_debug_set_gemm_trace(Tr.ptr), "trace")
f.yn = Ys[2]; f.out_op = Hs[2].ptr
check(lib.ns2vc_k_geglu(C.byref(f), PREC, None), "k_geglu"); sync()
check(lib.ns2vc_debug_set_gemm_trace(None), "trace")
t = Tr.to_numpy((nblk, 8, 16), dtype=np.uint64).astype(np.float64)
t0 = t[:, :, 0].min()
names = ["prologue", "wait tile (vmcnt)", "barrier", "issue + reads issued", "MFMAs + GEGLU chunk", "stores", "total"]
print(f"# geglu_kernel M={M}: {nblk} workgroups x 8 waves; entry spread {np.ptp(t[:, :, 0]):.0f}, last exit - first entry {t[:, :, 7].max() - t0:.0f} counter ticks (shader cycles)")
print("# median over the workgroups, per wave (waves w and w + 4 share a SIMD):")
print("  wave " + " ".join(f"{n:>22s}" for n in names))
for w in range(8):
    tw = t[:, w]
    tot = tw[:, 7] - tw[:, 0]
    cols = [tw[:, 1] - tw[:, 0]] + [tw[:, 2 + i] for i in range(5)] + [tot]
    print(f"  {w:4d} " + " ".join(f"{np.median(c):22.0f}" for c in cols))
hw = t[:, :, 8].astype(np.int64)
simd = (hw >> 4) & 3
print("# SIMD of waves 0..7 (HW_REG_HW_ID bits 4-5), first workgroups:", [simd[b].tolist() for b in range(4)])
print("# waves w and w + 4 on one SIMD in", int(np.sum(np.all(simd[:, :4] == simd[:, 4:], axis=1))), "of", nblk, "workgroups; waves 2k and 2k + 1 on one SIMD in",
      int(np.sum(np.all(simd[:, 0::2] == simd[:, 1::2], axis=1))))
