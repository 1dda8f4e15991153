"""GPU parity tests of the individual HIP kernels, driven through the C ABI
(ctypes + numpy only, no torch).  References are plain numpy float64 restatements
of the ATen ops the reference dispatches: conv1d / linear
(+GroupNorm/LayerNorm-apply prologues, GEGLU / bias / residual epilogues),
scaled_dot_product_attention, group_norm statistics, layer_norm statistics.

Tolerances (relative L2 on the whole output):
  fp32 "parity" kernels  : 2e-5   (exact-fp32 MFMA, only summation order differs)
  bf16 kernels           : 6e-3 vs a float64 reference fed the SAME bf16-rounded operands
"""
from __future__ import annotations

import ctypes as C
import zlib

import numpy as np
import pytest

from util import bf16_round, gather_rows, gelu_erf, rel_l2, silu

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 2e-5}      # operands are pre-rounded to the operand type, accumulation is fp32 in both modes


def _lib():
    from ns2vc_amd import _lib
    return _lib.load()


def _dev(a):
    from ns2vc_amd.engine import DevBuf
    return DevBuf.from_numpy(np.ascontiguousarray(a))


def _pack(W, prec):
    lib = _lib()
    from ns2vc_amd._lib import check
    W = np.ascontiguousarray(W, dtype=np.float32)
    p = C.c_void_p()
    check(lib.ns2vc_pack_weight(W.ctypes.data, W.shape[0], W.shape[1], prec, C.byref(p)), "pack_weight")
    return p


class OpBuf:
    """device buffer in the engine's operand type (bf16 for prec 1, fp32 for prec 0)"""

    def __init__(self, a, prec):
        from ns2vc_amd._lib import check
        a = np.ascontiguousarray(a, dtype=np.float32)
        self.prec, self.n, self.shape = prec, a.size, a.shape
        p = C.c_void_p()
        check(_lib().ns2vc_to_operand(a.ctypes.data, a.size, prec, C.byref(p)), "to_operand")
        self.ptr = p.value
        self.esz = 2 if prec == 1 else 4

    def read(self, shape=None):
        from ns2vc_amd._lib import check
        out = np.empty(self.n, dtype=np.float32)
        check(_lib().ns2vc_from_operand(self.ptr, self.n, self.prec, out.ctypes.data), "from_operand")
        return out.reshape(shape or self.shape)

    def __del__(self):
        try:
            _lib().ns2vc_dev_free(self.ptr)
        except Exception:
            pass


def rnd(a, prec):
    return bf16_round(np.asarray(a, dtype=np.float32)) if prec == 1 else np.asarray(a, dtype=np.float32)


def run_gemm(rng, prec, B, Tin, Tout, c0, c1, N, taps, tmode, bias_on, res_on, geglu, dual, tile=(0, 0, 0)):
    from ns2vc_amd._lib import GemmArgs, check
    from ns2vc_amd.engine import DevBuf, sync
    lib = _lib()
    Ct = c0 + c1
    K = taps * Ct
    M = B * Tout
    a0 = rnd(rng.standard_normal((B, Tin, c0)), prec)
    a1 = rnd(rng.standard_normal((B, Tin, c1)), prec) if c1 else None
    W = rnd(rng.standard_normal((N, K)) / np.sqrt(K), prec)
    bias = rng.standard_normal(N).astype(np.float32) if bias_on else None
    Nout = N // 2 if geglu else N
    res = rng.standard_normal((M, Nout)).astype(np.float32) if res_on else None
    A = (a0 if a1 is None else np.concatenate([a0, a1], axis=-1)).astype(np.float64)
    G = gather_rows(A, B, Tin, Tout, taps, tmode).reshape(M, K)
    ref = G @ W.astype(np.float64).T
    if bias is not None:
        ref = ref + bias
    if geglu:
        r3 = ref.reshape(M, N // 64, 2, 32)
        ref = (r3[:, :, 0, :] * gelu_erf(r3[:, :, 1, :])).reshape(M, N // 2)
    if res is not None:
        ref = ref + res

    d_a0, d_a1 = OpBuf(a0, prec), (OpBuf(a1, prec) if a1 is not None else None)
    d_w = _pack(W, prec)
    d_bias = _dev(bias) if bias is not None else None
    d_res = _dev(res) if res is not None else None
    d_out = DevBuf(M * Nout * 4)
    d_out.upload(np.full((M, Nout), np.nan, dtype=np.float32))
    d_oop = OpBuf(np.full((M, Nout), np.nan, dtype=np.float32), prec) if dual else None
    g = GemmArgs()
    g.a0 = d_a0.ptr; g.lda0 = c0; g.c0 = c0
    if d_a1 is not None:
        g.a1 = d_a1.ptr; g.lda1 = c1; g.c1 = c1
    g.B, g.Tin, g.Tout, g.M = B, Tin, Tout, M
    g.taps, g.tmode = taps, tmode
    g.w = d_w.value; g.K = K; g.N = N
    if d_bias is not None:
        g.bias = d_bias.ptr
    if d_res is not None:
        g.res = d_res.ptr; g.ldres = Nout
    g.geglu = int(geglu)
    g.out_f32 = d_out.ptr; g.ldo_f32 = Nout
    if d_oop is not None:
        g.out_op = d_oop.ptr; g.ldo_op = Nout
    check(lib.ns2vc_debug_set_gemm_tile(*tile), "set tile")
    try:
        check(lib.ns2vc_k_gemm(C.byref(g), prec, None), "k_gemm")
        sync()
    finally:
        lib.ns2vc_debug_set_gemm_tile(0, 0, 0)
    out = d_out.to_numpy((M, Nout))
    out_op = d_oop.read() if d_oop is not None else None
    lib.ns2vc_dev_free(d_w)
    return out, ref, out_op


GEMM_CASES = [
    # name, B, Tin, Tout, c0, c1, N, taps, tmode, bias, res, geglu, dual
    ("linear_plain", 2, 75, 75, 128, 0, 128, 1, 0, 1, 0, 0, 0),
    ("linear_res_tail_dual", 3, 41, 41, 256, 0, 192, 1, 0, 1, 1, 0, 1),
    ("linear_k_long", 2, 33, 33, 1024, 0, 256, 1, 0, 1, 1, 0, 0),
    ("linear_k_one_tile", 2, 50, 50, 64, 0, 128, 1, 0, 0, 0, 0, 0),
    ("conv3", 2, 37, 37, 128, 0, 128, 3, 0, 1, 0, 0, 0),
    ("conv3_concat_res", 2, 37, 37, 128, 64, 128, 3, 0, 1, 1, 0, 1),
    ("conv1_concat_shortcut", 2, 37, 37, 192, 128, 256, 1, 0, 1, 0, 0, 0),
    ("down2_odd", 2, 37, 19, 128, 0, 128, 3, 1, 1, 0, 0, 0),
    ("down2_even", 2, 38, 19, 128, 0, 128, 3, 1, 1, 0, 0, 0),
    ("up2_odd", 2, 19, 37, 128, 0, 128, 3, 2, 1, 0, 0, 0),
    ("up2_even", 2, 19, 38, 128, 0, 128, 3, 2, 1, 0, 0, 0),
    ("qkv_op_only_like", 2, 50, 50, 128, 0, 384, 1, 0, 1, 0, 0, 1),
    ("geglu", 2, 50, 50, 128, 0, 1024, 1, 0, 1, 0, 1, 1),
    ("temb_m_small", 3, 1, 1, 512, 0, 640, 1, 0, 1, 0, 0, 0),
]


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", GEMM_CASES, ids=[c[0] for c in GEMM_CASES])
def test_gemm_cases(case, prec, diag):
    name, *args = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    out, ref, out_op = run_gemm(rng, prec, *args)
    e = rel_l2(out, ref)
    diag(f"gemm {name} prec={prec} rel_l2={e:.3e} nan={int(np.isnan(out).sum())}")
    if not (e < TOL[prec]):
        bad = np.argwhere(~(np.abs(out - ref) <= 1e-2 + 1e-2 * np.abs(ref)))
        diag(f"  FAIL {name}: {len(bad)} bad of {out.size}; first {bad[:6].tolist()} rows_bad={sorted(set(bad[:, 0].tolist()))[:12]} cols_bad={sorted(set(bad[:, 1].tolist()))[:12]}")
    assert e < TOL[prec], (name, e)
    if out_op is not None:      # operand-typed copy == the fp32 result rounded to the operand type
        assert np.array_equal(out_op, rnd(out, prec)), name


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("tile", [(128, 128, 13), (64, 128, 13)], ids=lambda t: f"{t[0]}x{t[1]}s{t[2]}")
def test_gemm_cases_ksplit_kernel(tile, prec, diag):
    """Every feature case (taps, stride 2, upsample, concat, residual, GEGLU, dual outputs) through the 8-wave K-split kernel."""
    for name, *args in GEMM_CASES:
        if args[5] % 128:
            continue
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        out, ref, out_op = run_gemm(rng, prec, *args, tile=tile)
        e = rel_l2(out, ref)
        diag(f"gemm4 {name} tile={tile} prec={prec} rel_l2={e:.3e}")
        assert e < TOL[prec], (name, e)
        if out_op is not None:
            assert np.array_equal(out_op, rnd(out, prec)), name


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("tile", [(128, 128, 2), (128, 128, 3), (64, 128, 2), (64, 128, 3), (64, 128, 4), (128, 64, 2), (128, 64, 3),
                                  (128, 64, 4), (64, 64, 2), (64, 64, 3), (64, 64, 4),
                                  (128, 128, 1), (64, 128, 1), (128, 64, 1), (64, 64, 1),
                                  # stages 12..14 = the 8-wave K-split kernel (gemm4_kernel) with ring depth 2..4
                                  (128, 128, 12), (128, 128, 13), (128, 128, 14), (64, 128, 12), (64, 128, 13), (64, 128, 14)],
                         ids=lambda t: f"{t[0]}x{t[1]}s{t[2]}")
def test_gemm_every_tile(tile, prec, diag):
    rng = np.random.default_rng(tile[0] * 1000 + tile[1])
    # M = 3*167 = 501 rows (tail in every tile size), concat + conv3 + bias + residual, K = 3*192 (9 / 18 tiles)
    out, ref, _ = run_gemm(rng, prec, 3, 167, 167, 128, 64, 256, 3, 0, 1, 1, 0, 0, tile=tile)
    e = rel_l2(out, ref)
    diag(f"gemm tile={tile} prec={prec} rel_l2={e:.3e}")
    assert e < TOL[prec]
    # K of exactly 1, 2 and 3 tiles exercises the pipeline prologue / tail waits
    for kk in (1, 2, 3):
        out, ref, _ = run_gemm(rng, prec, 2, 90, 90, (64 if prec else 32) * kk, 0, 128, 1, 0, 1, 0, 0, 0, tile=tile)
        e = rel_l2(out, ref)
        diag(f"gemm tile={tile} prec={prec} ktiles={kk} rel_l2={e:.3e}")
        assert e < TOL[prec]
    if tile[1] == 128:
        out, ref, _ = run_gemm(rng, prec, 2, 90, 90, 256, 0, 512, 1, 0, 1, 0, 1, 0, tile=tile)
        e = rel_l2(out, ref)
        diag(f"gemm geglu tile={tile} prec={prec} rel_l2={e:.3e}")
        assert e < TOL[prec]


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
def test_gemm_epilogue_groupnorm_stats(prec, diag):
    """The epilogue's int64 fixed-point (sum, sumsq) per (batch item, 16-channel block) == numpy on the stored result,
    for every tile shape, with row tiles that straddle batch boundaries (T = 167 is not a multiple of 32)."""
    from ns2vc_amd._lib import GemmArgs, check
    from ns2vc_amd.engine import DevBuf, sync
    lib = _lib()
    rng = np.random.default_rng(5)
    B, T, c0, N = 3, 167, 128, 256
    for tile in [(0, 0, 0), (128, 128, 2), (64, 128, 2), (128, 64, 2), (64, 64, 2), (64, 128, 1), (64, 64, 1), (128, 128, 13), (64, 128, 13)]:
        a0 = rnd(rng.standard_normal((B, T, c0)), prec)
        W = rnd(rng.standard_normal((N, 3 * c0)) / np.sqrt(3 * c0), prec)
        bias = rng.standard_normal(N).astype(np.float32)
        res = rng.standard_normal((B * T, N)).astype(np.float32)
        d_a, d_w, d_b, d_r = OpBuf(a0, prec), _pack(W, prec), _dev(bias), _dev(res)
        d_o = DevBuf(B * T * N * 4)
        d_s = DevBuf.from_numpy(np.zeros((B, N // 16, 2), dtype=np.int64))
        g = GemmArgs()
        g.a0 = d_a.ptr; g.lda0 = c0; g.c0 = c0
        g.B, g.Tin, g.Tout, g.M = B, T, T, B * T
        g.taps, g.tmode = 3, 0
        g.w = d_w.value; g.K = 3 * c0; g.N = N
        g.bias = d_b.ptr; g.res = d_r.ptr; g.ldres = N
        g.out_f32 = d_o.ptr; g.ldo_f32 = N
        g.stats = d_s.ptr
        check(lib.ns2vc_debug_set_gemm_tile(*tile), "tile")
        try:
            check(lib.ns2vc_k_gemm(C.byref(g), prec, None), "k_gemm")
            sync()
        finally:
            lib.ns2vc_debug_set_gemm_tile(0, 0, 0)
        out = d_o.to_numpy((B, T, N)).astype(np.float64)
        st = d_s.to_numpy((B, N // 16, 2), dtype=np.int64).astype(np.float64)
        blk = out.reshape(B, T, N // 16, 16)
        ref_s, ref_q = blk.sum(axis=(1, 3)), (blk ** 2).sum(axis=(1, 3))
        e_s = np.abs(st[..., 0] / 2 ** 28 - ref_s).max() / np.abs(ref_s).max()
        e_q = np.abs(st[..., 1] / 2 ** 16 - ref_q).max() / np.abs(ref_q).max()
        diag(f"gemm epilogue stats tile={tile} prec={prec}: sum {e_s:.2e} sumsq {e_q:.2e}")
        assert e_s < 1e-5 and e_q < 1e-5
        lib.ns2vc_dev_free(d_w)


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("tile", [(0, 0, 0), (64, 128, 2), (64, 64, 2), (64, 128, 13), (128, 128, 13)], ids=lambda t: f"{t[0]}x{t[1]}s{t[2]}")
def test_gemm_layernorm_by_linearity(tile, prec, diag):
    """LayerNorm(y) @ W'^T without a normalisation pass (attention.py:83,102,118): the producer GEMM leaves (sum, sumsq)
    per row and 64-column slice and an operand copy of y; the consumer GEMM reads the raw copy and applies
    rstd * (acc - mean * rowsum(W')) + b in its epilogue.  Checked against numpy LayerNorm -> matmul in fp64,
    plain and GEGLU consumers, rows with a large mean (cancellation) included."""
    from ns2vc_amd._lib import GemmArgs, check
    from ns2vc_amd.engine import DevBuf, sync
    lib = _lib()
    rng = np.random.default_rng(11)
    B, T, D = 3, 83, 256                      # M = 249: row tail in every tile
    M = B * T
    # ---- producer: y = a @ W1^T + b1 + res  (fp32 + operand copy + row statistics)
    a = rnd(rng.standard_normal((M, D)), prec)
    W1 = rnd(rng.standard_normal((D, D)) / np.sqrt(D), prec)
    b1 = rng.standard_normal(D).astype(np.float32)
    res = (rng.standard_normal((M, D)) + 3.0 * rng.standard_normal((M, 1))).astype(np.float32)   # per-row offsets: |mean| up to ~3 sigma
    d_a, d_w1, d_b1, d_res = OpBuf(a, prec), _pack(W1, prec), _dev(b1), _dev(res)
    d_y = DevBuf(M * D * 4)
    d_yop = DevBuf(M * D * (2 if prec else 4))
    d_rs = DevBuf.from_numpy(np.full((M, D // 64, 2), np.nan, dtype=np.float32))     # every slot must be written
    g = GemmArgs()
    g.a0 = d_a.ptr; g.lda0 = D; g.c0 = D
    g.B, g.Tin, g.Tout, g.M = B, T, T, M
    g.taps, g.tmode = 1, 0
    g.w = d_w1.value; g.K = D; g.N = D; g.bias = d_b1.ptr
    g.res = d_res.ptr; g.ldres = D
    g.out_f32 = d_y.ptr; g.ldo_f32 = D
    g.out_op = d_yop.ptr; g.ldo_op = D
    g.rowstats = d_rs.ptr
    try:
        check(lib.ns2vc_debug_set_gemm_tile(*(tile if tile[1] != 64 else (0, 0, 0))), "tile")   # producers need 64-column wave tiles
        check(lib.ns2vc_k_gemm(C.byref(g), prec, None), "producer gemm")
        sync()
        check(lib.ns2vc_debug_set_gemm_tile(*tile), "tile")
        y = d_y.to_numpy((M, D)).astype(np.float64)
        st = d_rs.to_numpy((M, D // 64, 2)).astype(np.float64)
        ys = y.reshape(M, D // 64, 64)
        e_s = np.abs(st[..., 0] - ys.sum(2)).max() / np.abs(ys.sum(2)).max()
        e_q = np.abs(st[..., 1] - (ys ** 2).sum(2)).max() / (ys ** 2).sum(2).max()
        diag(f"ln-linear producer tile={tile} prec={prec}: slice sums {e_s:.2e} slice sumsq {e_q:.2e}")
        assert e_s < 1e-5 and e_q < 1e-5
        # ---- consumers
        mu = y.mean(1, keepdims=True)
        yn = (y - mu) / np.sqrt(y.var(1, keepdims=True) + 1e-5)
        for geglu, N in ((0, 384), (1, 512)):
            W2 = rnd(rng.standard_normal((N, D)) / np.sqrt(D), prec)
            b2 = rng.standard_normal(N).astype(np.float32)
            d_w2, d_b2 = _pack(W2, prec), _dev(b2)
            ws = C.c_void_p()
            W2c = np.ascontiguousarray(W2, dtype=np.float32)
            check(lib.ns2vc_weight_rowsum(W2c.ctypes.data, N, D, prec, C.byref(ws)), "rowsum")
            Nout = N // 2 if geglu else N
            d_o = DevBuf(M * Nout * 4)
            g2 = GemmArgs()
            g2.a0 = d_yop.ptr; g2.lda0 = D; g2.c0 = D
            g2.B, g2.Tin, g2.Tout, g2.M = B, T, T, M
            g2.taps, g2.tmode = 1, 0
            g2.w = d_w2.value; g2.K = D; g2.N = N; g2.bias = d_b2.ptr
            g2.geglu = geglu
            g2.out_f32 = d_o.ptr; g2.ldo_f32 = Nout
            g2.ln_stats = d_rs.ptr; g2.ln_wsum = ws.value; g2.ln_eps = 1e-5; g2.ln_dim = D
            if geglu and tile[1] not in (0, 128):
                continue
            check(lib.ns2vc_k_gemm(C.byref(g2), prec, None), "consumer gemm")
            sync()
            out = d_o.to_numpy((M, Nout)).astype(np.float64)
            pre = yn @ W2.astype(np.float64).T + b2
            if geglu:   # packed layout: groups of (32 value | 32 gate) columns
                from scipy.special import erf
                pg = pre.reshape(M, N // 64, 2, 32)
                ref = (pg[:, :, 0] * 0.5 * pg[:, :, 1] * (1.0 + erf(pg[:, :, 1] / np.sqrt(2.0)))).reshape(M, Nout)
            else:
                ref = pre
            e = rel_l2(out, ref)
            diag(f"ln-linear consumer tile={tile} prec={prec} geglu={geglu}: rel_l2 {e:.3e}")
            # fp32: exact up to rounding; bf16: the raw operand copy is rounded BEFORE normalisation (2^-9 of |y|, not of |y - mean|)
            assert e < (2e-5 if prec == 0 else 1.5e-2)
            lib.ns2vc_dev_free(d_w2); lib.ns2vc_dev_free(ws)
    finally:
        lib.ns2vc_debug_set_gemm_tile(0, 0, 0)
    lib.ns2vc_dev_free(d_w1)


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", [("plain", 128, 0, 128, False, False, False), ("concat_raw_shortcut", 128, 128, 256, True, False, True),
                                  ("temb_res", 256, 0, 256, False, True, False), ("wide_c384", 384, 0, 128, False, True, False),
                                  ("c512", 512, 0, 128, False, False, False)], ids=lambda c: c[0])
def test_conv3_fused_groupnorm(case, prec, diag):
    """conv3(act(GroupNorm(x) [* (1 + scale) + shift])) in ONE launch (resnet.py:591-641): the kernel reads the fp32 rows,
    normalises from the producers' int64 statistics, keeps the operand panel in LDS for all three taps.  Sequence
    ends inside a 64-row tile (T = 70), concat input, raw operand copy, fused 1x1 shortcut segment, residual, result
    statistics -- against numpy in fp64."""
    from ns2vc_amd._lib import ConvGnArgs, check
    from ns2vc_amd.engine import DevBuf, sync
    name, c0, c1, N, shortcut, use_temb, want_raw = case
    if prec == 0 and c0 + c1 > 320:
        pytest.skip("fp32 panel of > 320 channels does not fit LDS: the engine falls back to gn_apply + GEMM there")
    lib = _lib()
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    B, T, G = 3, 70, 8
    C_ = c0 + c1
    M = B * T
    x0 = (rng.standard_normal((B, T, c0)) * 1.5 + 0.3).astype(np.float32)
    x1 = (rng.standard_normal((B, T, c1)) - 0.2).astype(np.float32) if c1 else None
    x = np.concatenate([x0, x1], axis=-1) if c1 else x0
    gamma = (1.0 + 0.1 * rng.standard_normal(C_)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C_)).astype(np.float32)
    temb = (0.2 * rng.standard_normal((B, 2 * C_ + 8))).astype(np.float32) if use_temb else None
    c2 = 192 if shortcut else 0
    K = 3 * C_ + c2
    W = rnd(rng.standard_normal((N, K)) / np.sqrt(K), prec)
    bias = rng.standard_normal(N).astype(np.float32)
    res = None if shortcut else rng.standard_normal((M, N)).astype(np.float32)
    a2 = rnd(rng.standard_normal((B, T, c2)), prec) if shortcut else None

    def stats_of(t):          # what a producing GEMM's epilogue leaves: int64 fixed point per (batch item, 16-channel block)
        blk = t.astype(np.float64).reshape(B, T, t.shape[-1] // 16, 16)
        return np.stack([np.rint(blk.sum(axis=(1, 3)) * 2 ** 28), np.rint((blk ** 2).sum(axis=(1, 3)) * 2 ** 16)], axis=-1).astype(np.int64)

    x64 = x.astype(np.float64)
    xg = x64.reshape(B, T, G, C_ // G)
    mean, var = xg.mean(axis=(1, 3), keepdims=True), xg.var(axis=(1, 3), keepdims=True)
    y = ((xg - mean) / np.sqrt(var + 1e-5)).reshape(B, T, C_) * gamma + beta
    if use_temb:
        y = y * (1.0 + temb[:, None, 4:4 + C_]) + temb[:, None, 4 + C_:4 + 2 * C_]
    y = y / (1.0 + np.exp(-y))
    yb = rnd(y, prec).astype(np.float64)
    ypad = np.pad(yb, ((0, 0), (1, 1), (0, 0)))
    Wd = W.astype(np.float64)
    ref = np.zeros((B, T, N))
    for tap in range(3):
        ref += ypad[:, tap:tap + T, :] @ Wd[:, tap * C_:(tap + 1) * C_].T
    if shortcut:
        ref += a2.astype(np.float64) @ Wd[:, 3 * C_:].T
    ref = ref.reshape(M, N) + bias
    if res is not None:
        ref += res

    a = ConvGnArgs()
    d_x0, d_x1 = _dev(x0), (_dev(x1) if c1 else None)
    d_s0, d_s1 = _dev(stats_of(x0)), (_dev(stats_of(x1)) if c1 else None)
    d_g, d_b, d_w, d_bias = _dev(gamma), _dev(beta), _pack(W, prec), _dev(bias)
    d_o = DevBuf(M * N * 4)
    d_st = DevBuf.from_numpy(np.zeros((B, N // 16, 2), dtype=np.int64))
    a.x0, a.ldx0 = d_x0.ptr, c0
    if c1:
        a.x1, a.ldx1 = d_x1.ptr, c1
        a.st1 = d_s1.ptr
    a.st0 = d_s0.ptr
    a.gamma, a.beta, a.groups, a.eps, a.silu = d_g.ptr, d_b.ptr, G, 1e-5, 1
    if use_temb:
        d_t = _dev(temb)
        a.temb, a.ldtemb, a.temb_off = d_t.ptr, temb.shape[1], 4
    d_raw = None
    if want_raw:
        d_raw = OpBuf(np.full((M, C_), np.nan, np.float32), prec)
        a.raw_op = d_raw.ptr
    g = a.g
    g.c0, g.c1 = c0, c1
    g.B, g.Tin, g.Tout, g.M = B, T, T, M
    g.taps, g.tmode = 3, 0
    g.w, g.K, g.N, g.bias = d_w.value, K, N, d_bias.ptr
    if shortcut:
        d_a2 = OpBuf(a2, prec)
        g.a2, g.lda2, g.c2 = d_a2.ptr, c2, c2
    else:
        d_res = _dev(res)
        g.res, g.ldres = d_res.ptr, N
    g.out_f32, g.ldo_f32 = d_o.ptr, N
    g.stats = d_st.ptr
    check(lib.ns2vc_k_convgn(C.byref(a), prec, None), "k_convgn")
    sync()
    out = d_o.to_numpy((M, N))
    e = rel_l2(out, ref)
    diag(f"conv3+groupnorm {name} prec={prec}: rel_l2 {e:.3e} nan={int(np.isnan(out).sum())}")
    if not e < (3e-5 if prec == 0 else 6e-3):
        err = np.abs(out - ref)
        diag(f"  FAIL rows {sorted(set(np.argwhere(err > 0.05)[:, 0].tolist()))[:24]} cols {sorted(set(np.argwhere(err > 0.05)[:, 1].tolist()))[:12]}")
    assert e < (3e-5 if prec == 0 else 6e-3), e          # bf16: the normalised activations are rounded to bf16 (as gn_apply does)
    st = d_st.to_numpy((B, N // 16, 2), dtype=np.int64).astype(np.float64)
    ob = out.astype(np.float64).reshape(B, T, N // 16, 16)
    assert np.abs(st[..., 0] / 2 ** 28 - ob.sum(axis=(1, 3))).max() / np.abs(ob.sum(axis=(1, 3))).max() < 1e-5
    if want_raw:
        assert np.array_equal(d_raw.read((M, C_)), rnd(x.reshape(M, C_), prec))
    lib.ns2vc_dev_free(d_w)


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
def test_gemm_fused_shortcut_segment(prec, diag):
    """conv3(hn) + conv1x1(x) in one launch: K = 3*c0 + c2 with the second segment on another operand tensor."""
    from ns2vc_amd._lib import GemmArgs, check
    from ns2vc_amd.engine import DevBuf, sync
    lib = _lib()
    rng = np.random.default_rng(9)
    for (B, T, c0, c2, N) in [(2, 37, 128, 192, 128), (3, 70, 256, 640, 256)]:
        M = B * T
        hn = rnd(rng.standard_normal((B, T, c0)), prec)
        x = rnd(rng.standard_normal((B, T, c2)), prec)
        W = rnd(rng.standard_normal((N, 3 * c0 + c2)) / np.sqrt(3 * c0 + c2), prec)
        bias = rng.standard_normal(N).astype(np.float32)
        G = gather_rows(hn.astype(np.float64), B, T, T, 3, 0).reshape(M, 3 * c0)
        ref = G @ W[:, :3 * c0].astype(np.float64).T + x.reshape(M, c2).astype(np.float64) @ W[:, 3 * c0:].astype(np.float64).T + bias
        d_h, d_x, d_w, d_b = OpBuf(hn, prec), OpBuf(x, prec), _pack(W, prec), _dev(bias)
        d_o = DevBuf(M * N * 4)
        g = GemmArgs()
        g.a0 = d_h.ptr; g.lda0 = c0; g.c0 = c0
        g.a2 = d_x.ptr; g.lda2 = c2; g.c2 = c2
        g.B, g.Tin, g.Tout, g.M = B, T, T, M
        g.taps, g.tmode = 3, 0
        g.w = d_w.value; g.K = 3 * c0 + c2; g.N = N; g.bias = d_b.ptr
        g.out_f32 = d_o.ptr; g.ldo_f32 = N
        check(lib.ns2vc_k_gemm(C.byref(g), prec, None), "k_gemm")
        sync()
        e = rel_l2(d_o.to_numpy((M, N)), ref)
        diag(f"gemm fused shortcut {(B, T, c0, c2, N)} prec={prec}: {e:.3e}")
        assert e < TOL[prec]
        lib.ns2vc_dev_free(d_w)


def test_gemm_heuristic_large(diag):
    """A level-0 sized problem (M = 4*938) goes through the tile heuristic."""
    rng = np.random.default_rng(7)
    for prec in (0, 1):
        out, ref, _ = run_gemm(rng, prec, 4, 938, 938, 128, 0, 128, 3, 0, 1, 1, 0, 0)
        e = rel_l2(out, ref)
        diag(f"gemm level0 prec={prec} rel_l2={e:.3e}")
        assert e < TOL[prec]


def _chain_stream(mats):
    from ns2vc_amd._lib import check
    arrs = [np.ascontiguousarray(m, dtype=np.float32) for m in mats]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    Ns = (C.c_int * len(arrs))(*[a.shape[0] for a in arrs])
    Ks = (C.c_int * len(arrs))(*[a.shape[1] for a in arrs])
    out = C.c_void_p()
    check(_lib().ns2vc_pack_chain_stream(ptrs, Ns, Ks, len(arrs), C.byref(out)), "pack_chain_stream")
    return out


@pytest.mark.parametrize("shape", [(130, 128, 384, True), (333, 256, 256, True), (100, 384, 1152, False), (77, 512, 512, True),
                                   (30016 // 8, 128, 384, False)], ids=str)
def test_chain_linear_layernorm_linear(shape, diag):
    """Fused row chain (bf16): y = A W1^T + b1 (+res); out2 = LayerNorm(y) W2^T + b2 — vs float64 numpy on the same
    bf16-rounded operands.  y is fp32; out2 differs from the reference only by the bf16 rounding of LN(y) and of itself."""
    from ns2vc_amd._lib import check
    from ns2vc_amd.engine import DevBuf, sync
    lib = _lib()
    M, D, N2, with_res = shape
    rng = np.random.default_rng(M + D)
    a = bf16_round(rng.standard_normal((M, D)).astype(np.float32))
    W1 = bf16_round((rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32))
    W2 = bf16_round((rng.standard_normal((N2, D)) / np.sqrt(D)).astype(np.float32))
    b1, b2 = rng.standard_normal(D).astype(np.float32), rng.standard_normal(N2).astype(np.float32)
    res = (rng.standard_normal((M, D)) * 1.5 + 0.5).astype(np.float32) if with_res else None
    y_ref = a.astype(np.float64) @ W1.astype(np.float64).T + b1 + (res if res is not None else 0.0)
    n = (y_ref - y_ref.mean(-1, keepdims=True)) / np.sqrt(y_ref.var(-1, keepdims=True) + 1e-5)
    o_ref = bf16_round(n.astype(np.float32)).astype(np.float64) @ W2.astype(np.float64).T + b2
    ws = _chain_stream([W1, W2])
    d_a, d_b1, d_b2 = OpBuf(a, 1), _dev(b1), _dev(b2)
    d_y = DevBuf.from_numpy(res.copy()) if with_res else DevBuf(M * D * 4)      # in place: res aliases y like the engine does
    d_o = OpBuf(np.full((M, N2), np.nan, np.float32), 1)
    check(lib.ns2vc_k_chain_ab(d_a.ptr, M, D, ws, d_b1.ptr, d_y.ptr if with_res else None, d_y.ptr, 1e-5, d_b2.ptr, d_o.ptr, N2, None), "chain_ab")
    sync()
    y = d_y.to_numpy((M, D))
    o = d_o.read()
    e_y, e_o = rel_l2(y, y_ref), rel_l2(o, o_ref)
    diag(f"chain_ab {shape}: y {e_y:.3e} out2 {e_o:.3e}")
    assert e_y < 2e-6 and e_o < 6e-3
    lib.ns2vc_dev_free(ws)


# ---------------------------------------------------------------------------------------
def ref_attention(q, k, v, bias, H, prec):
    B, Lq, D = q.shape
    Lk = k.shape[1]
    hd = D // H
    qh = q.reshape(B, Lq, H, hd).transpose(0, 2, 1, 3).astype(np.float64)
    kh = k.reshape(B, Lk, H, hd).transpose(0, 2, 1, 3).astype(np.float64)
    vh = v.reshape(B, Lk, H, hd).transpose(0, 2, 1, 3).astype(np.float64)
    s = qh @ kh.transpose(0, 1, 3, 2) / np.sqrt(hd)
    if bias is not None:
        s = s + bias[:, None, None, :]
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = p @ vh
    return o.transpose(0, 2, 1, 3).reshape(B, Lq, D)


ATTN_CASES = [
    # name, B, H, hd, Lq, Lk, bias, packed_qkv
    ("self_hd16", 2, 8, 16, 150, 150, False, True),
    ("self_hd32", 2, 8, 32, 75, 75, False, True),
    ("self_hd48", 1, 8, 48, 130, 130, False, True),
    ("self_hd64", 2, 4, 64, 64, 64, False, True),
    ("self_tiny", 1, 2, 16, 5, 5, False, True),
    ("cross_hd16_mask", 2, 8, 16, 150, 69, True, False),
    ("cross_hd32_mask", 2, 8, 32, 70, 130, True, False),
    ("cross_hd48_mask", 2, 8, 48, 33, 21, True, False),
    ("cross_hd64_mask", 2, 8, 64, 40, 469, True, False),
]


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention(case, prec, diag):
    from ns2vc_amd._lib import AttnArgs, check
    from ns2vc_amd.engine import DevBuf, sync
    name, B, H, hd, Lq, Lk, use_bias, packed = case
    lib = _lib()
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    D = H * hd
    q = rng.standard_normal((B, Lq, D)).astype(np.float32)
    k = rng.standard_normal((B, Lk, D)).astype(np.float32)
    v = rng.standard_normal((B, Lk, D)).astype(np.float32)
    bias = None
    if use_bias:
        keep = rng.random((B, Lk)) > 0.3
        keep[:, 0] = True
        keep[0, Lk // 2:] = False          # a padded tail, like a ragged prompt batch
        bias = np.where(keep, 0.0, -10000.0).astype(np.float32)
    if prec == 1:
        ref = ref_attention(bf16_round(q), bf16_round(k), bf16_round(v), bias, H, prec)
    else:
        ref = ref_attention(q, k, v, bias, H, prec)
    a = AttnArgs()
    esz = 2 if prec == 1 else 4
    if packed:   # q|k|v interleaved per row, as the fused QKV GEMM writes them
        d_qkv = OpBuf(np.concatenate([q, k, v], axis=-1), prec)
        a.q, a.k, a.v = d_qkv.ptr, d_qkv.ptr + D * esz, d_qkv.ptr + 2 * D * esz
        a.ldq = a.ldk = a.ldv = 3 * D
    else:        # k|v side by side with extra columns around, as the hoisted cross K/V buffer
        pad = 64
        kv = np.concatenate([np.zeros((B, Lk, pad), np.float32), k, v, np.zeros((B, Lk, pad), np.float32)], axis=-1)
        d_q, d_kv = OpBuf(q, prec), OpBuf(kv, prec)
        a.q, a.k, a.v = d_q.ptr, d_kv.ptr + pad * esz, d_kv.ptr + (pad + D) * esz
        a.ldq, a.ldk, a.ldv = D, 2 * D + 2 * pad, 2 * D + 2 * pad
    a.B, a.H, a.Lq, a.Lk = B, H, Lq, Lk
    d_bias = _dev(bias) if bias is not None else None
    if d_bias is not None:
        a.bias = d_bias.ptr
    a.scale = 1.0 / np.sqrt(hd)
    d_out = OpBuf(np.full((B, Lq, D), np.nan, dtype=np.float32), prec)
    a.out, a.ldo = d_out.ptr, D
    check(lib.ns2vc_k_attention(C.byref(a), hd, prec, None), "k_attention")
    sync()
    out = d_out.read((B, Lq, D))
    e = rel_l2(out, ref)
    diag(f"attn {name} prec={prec} rel_l2={e:.3e} nan={int(np.isnan(out).sum())}")
    tol = 2e-5 if prec == 0 else 1.5e-2      # bf16: P is rounded to bf16 before the PV MFMA
    if not e < tol:
        err = np.abs(out - ref).reshape(B, Lq, H, hd)
        diag(f"  FAIL {name}: per-head max err {err.max(axis=(0, 1, 3)).round(4).tolist()} per-d max {err.max(axis=(0, 1, 2)).round(3).tolist()[:16]}")
        diag(f"  per-q (b0,h0) {err[0, :, 0, :].max(-1).round(3).tolist()[:40]}")
    assert e < tol, (name, e)


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
def test_attention_reference_shift_extremes(prec, diag):
    """The kernel keeps a per-query softmax REFERENCE instead of the running max (moved only when a score exceeds it by
    2^12): large logits, a sharply growing maximum, and a first key tile that is entirely masked must all stay exact."""
    from ns2vc_amd._lib import AttnArgs, check
    from ns2vc_amd.engine import sync
    lib = _lib()
    rng = np.random.default_rng(77)
    B, H, hd, Lq, Lk = 2, 4, 32, 96, 300
    D = H * hd
    q = (4.0 * rng.standard_normal((B, Lq, D))).astype(np.float32)          # logits of +-60 and more
    k = (3.0 * rng.standard_normal((B, Lk, D))).astype(np.float32)
    k *= np.linspace(0.2, 2.5, Lk, dtype=np.float32)[None, :, None]          # later keys produce ever larger scores
    v = rng.standard_normal((B, Lk, D)).astype(np.float32)
    keep = np.ones((B, Lk), dtype=bool)
    keep[0, :70] = False                                                       # the whole first 64-key tile is masked
    keep[1, 200:] = False
    bias = np.where(keep, 0.0, -10000.0).astype(np.float32)
    qr, kr, vr = (rnd(t, prec) for t in (q, k, v))
    ref = ref_attention(qr, kr, vr, bias, H, prec)
    esz = 2 if prec == 1 else 4
    a = AttnArgs()
    d_q, d_kv = OpBuf(q, prec), OpBuf(np.concatenate([k, v], axis=-1), prec)
    a.q, a.k, a.v = d_q.ptr, d_kv.ptr, d_kv.ptr + D * esz
    a.ldq, a.ldk, a.ldv = D, 2 * D, 2 * D
    a.B, a.H, a.Lq, a.Lk = B, H, Lq, Lk
    d_bias = _dev(bias)
    a.bias = d_bias.ptr
    a.scale = 1.0 / np.sqrt(hd)
    d_out = OpBuf(np.full((B, Lq, D), np.nan, dtype=np.float32), prec)
    a.out, a.ldo = d_out.ptr, D
    check(lib.ns2vc_k_attention(C.byref(a), hd, prec, None), "k_attention")
    sync()
    out = d_out.read((B, Lq, D))
    e = rel_l2(out, ref)
    diag(f"attn reference-shift extremes prec={prec}: rel_l2={e:.3e} nan={int(np.isnan(out).sum())} inf={int(np.isinf(out).sum())}")
    assert np.isfinite(out).all()
    assert e < (3e-5 if prec == 0 else 3e-2), e        # bf16: near one-hot softmax over bf16-rounded P


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("ln", [False, True], ids=["plain", "ln_linear"])
@pytest.mark.parametrize("hd,Lq,Lk", [(16, 150, 69), (32, 70, 130), (48, 33, 21), (64, 40, 200)])
def test_attention_fused_query_projection(hd, Lq, Lk, ln, prec, diag):
    """Cross-attention whose to_q projection (attention_processor.py:1013) runs inside the attention kernel:
    Q = x @ Wq^T + bq, optionally with x the RAW input of a LayerNorm (statistics pairs + rowsum fix-up), against
    numpy LayerNorm -> matmul -> masked softmax attention in fp64."""
    from ns2vc_amd._lib import AttnArgs, check
    from ns2vc_amd.engine import DevBuf, sync
    lib = _lib()
    B, H = 2, 8
    D = H * hd
    rng = np.random.default_rng(hd * 1000 + Lq)
    x = (rng.standard_normal((B, Lq, D)) + (1.5 * rng.standard_normal((B, Lq, 1)) if ln else 0.0)).astype(np.float32)
    Wq = rnd(rng.standard_normal((D, D)) / np.sqrt(D), prec)
    bq = (0.3 * rng.standard_normal(D)).astype(np.float32)
    k = rng.standard_normal((B, Lk, D)).astype(np.float32)
    v = rng.standard_normal((B, Lk, D)).astype(np.float32)
    keep = rng.random((B, Lk)) > 0.3
    keep[:, 0] = True
    bias = np.where(keep, 0.0, -10000.0).astype(np.float32)
    xr = rnd(x, prec).astype(np.float64)
    if ln:
        x64 = x.astype(np.float64)
        xin = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5)
    else:
        xin = xr
    q = xin @ Wq.astype(np.float64).T + bq
    ref = ref_attention(rnd(q, prec), rnd(k, prec), rnd(v, prec), bias, H, prec)
    esz = 2 if prec == 1 else 4
    a = AttnArgs()
    d_x, d_w = OpBuf(x, prec), _pack(Wq, prec)
    d_bq = _dev(bq)
    kv = np.concatenate([k, v], axis=-1)
    d_kv = OpBuf(kv, prec)
    a.k, a.v = d_kv.ptr, d_kv.ptr + D * esz
    a.ldk = a.ldv = 2 * D
    a.B, a.H, a.Lq, a.Lk = B, H, Lq, Lk
    d_bias = _dev(bias)
    a.bias = d_bias.ptr
    a.scale = 1.0 / np.sqrt(hd)
    d_out = OpBuf(np.full((B, Lq, D), np.nan, dtype=np.float32), prec)
    a.out, a.ldo = d_out.ptr, D
    a.xq, a.ldx, a.xdim, a.wq, a.bq = d_x.ptr, D, D, d_w.value, d_bq.ptr
    keepalive = []
    if ln:
        x64 = x.astype(np.float64).reshape(B * Lq, D // 64, 64)
        st = np.stack([x64.sum(-1), (x64 ** 2).sum(-1)], axis=-1).astype(np.float32)
        d_st = _dev(st)
        ws = C.c_void_p()
        Wc = np.ascontiguousarray(Wq, dtype=np.float32)
        check(lib.ns2vc_weight_rowsum(Wc.ctypes.data, D, D, prec, C.byref(ws)), "rowsum")
        a.ln_stats, a.ln_wsum, a.ln_eps, a.ln_dim = d_st.ptr, ws.value, 1e-5, D
        keepalive += [d_st]
    check(lib.ns2vc_k_attention(C.byref(a), hd, prec, None), "k_attention (fused to_q)")
    sync()
    out = d_out.read((B, Lq, D))
    e = rel_l2(out, ref)
    diag(f"attn fused-to_q hd={hd} ln={ln} prec={prec} rel_l2={e:.3e} nan={int(np.isnan(out).sum())}")
    assert e < (3e-5 if prec == 0 else 2e-2), e
    lib.ns2vc_dev_free(d_w)
    if ln:
        lib.ns2vc_dev_free(ws)


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 37, 128, 0), (2, 90, 512, 384), (3, 200, 384, 256), (1, 5, 128, 128), (2, 938, 128, 0)], ids=str)
def test_groupnorm(shape, prec, diag):
    """group_norm (+ the resnet's time scale/shift, + SiLU) over a two-source concat whose groups straddle the seam."""
    from ns2vc_amd._lib import check
    lib = _lib()
    B, T, c0, c1 = shape
    G = 8
    C_ = c0 + c1
    rng = np.random.default_rng(B * 1000 + T)
    a0 = (rng.standard_normal((B, T, c0)) * 2 + 0.7).astype(np.float32)
    a1 = (rng.standard_normal((B, T, c1)) - 0.4).astype(np.float32) if c1 else None
    gamma = (1 + 0.1 * rng.standard_normal(C_)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C_)).astype(np.float32)
    temb = rng.standard_normal((B, 2 * C_ + 64)).astype(np.float32) * 0.2
    off = 64
    A = (a0 if a1 is None else np.concatenate([a0, a1], -1)).astype(np.float64)
    Ag = A.reshape(B, T, G, C_ // G)
    mean = Ag.mean(axis=(1, 3), keepdims=True)
    var = Ag.var(axis=(1, 3), keepdims=True)
    gn = ((Ag - mean) / np.sqrt(var + 1e-5)).reshape(B, T, C_) * gamma + beta
    d_a0, d_a1 = _dev(a0), (_dev(a1) if a1 is not None else None)
    d_g, d_b, d_t = _dev(gamma), _dev(beta), _dev(temb)
    for with_t, silu_on in ((False, 1), (True, 1), (False, 0)):
        ref = gn
        if with_t:
            ref = ref * (1 + temb[:, None, off:off + C_]) + temb[:, None, off + C_:off + 2 * C_]
        if silu_on:
            ref = silu(ref)
        d_o = OpBuf(np.full((B, T, C_), np.nan, np.float32), prec)
        d_r = OpBuf(np.full((B, T, C_), np.nan, np.float32), prec)
        check(lib.ns2vc_k_groupnorm(d_a0.ptr, c0, c0, d_a1.ptr if d_a1 else None, c1, c1, B, T, G, 1e-5, d_g.ptr, d_b.ptr,
                                    d_t.ptr if with_t else None, temb.shape[1], off, silu_on, d_o.ptr, d_r.ptr, prec, None), "groupnorm")
        e1, e2 = rel_l2(d_o.read(), ref), rel_l2(d_r.read(), A)
        diag(f"groupnorm {shape} prec={prec} temb={with_t} silu={silu_on}: out {e1:.2e} raw {e2:.2e}")
        assert e1 < (5e-6 if prec == 0 else 4e-3) and e2 < (1e-7 if prec == 0 else 4e-3)


@pytest.mark.parametrize("prec", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(77, 128), (300, 512), (5, 384), (1000, 256)], ids=str)
def test_layernorm_apply(shape, prec, diag):
    from ns2vc_amd._lib import check
    from ns2vc_amd.engine import sync
    lib = _lib()
    M, C_ = shape
    rng = np.random.default_rng(M)
    x = (rng.standard_normal((M, C_)) * 1.5 + 0.3).astype(np.float32)
    d_x = _dev(x)
    d_o = OpBuf(np.full((M, C_), np.nan, np.float32), prec)
    check(lib.ns2vc_k_layernorm_apply(d_x.ptr, C_, M, C_, 1e-5, d_o.ptr, prec, None), "ln_apply")
    sync()
    xd = x.astype(np.float64)
    ref = (xd - xd.mean(-1, keepdims=True)) / np.sqrt(xd.var(-1, keepdims=True) + 1e-5)
    e = rel_l2(d_o.read(), ref)
    diag(f"layernorm_apply {shape} prec={prec} rel_l2={e:.3e}")
    assert e < (2e-6 if prec == 0 else 4e-3)


def test_layout_roundtrip(diag):
    from ns2vc_amd._lib import check
    from ns2vc_amd.engine import DevBuf, sync
    lib = _lib()
    rng = np.random.default_rng(3)
    for (B, C_, T, cpad) in [(2, 100, 37, 128), (3, 256, 188, 256), (1, 100, 938, 128)]:
        x = rng.standard_normal((B, C_, T)).astype(np.float32)
        d_x, d_y, d_z = _dev(x), DevBuf(B * T * cpad * 4), DevBuf(B * C_ * T * 4)
        d_y.upload(np.full((B, T, cpad), np.nan, np.float32))
        check(lib.ns2vc_k_nct_to_btc(d_x.ptr, C_, T, B, d_y.ptr, cpad, cpad, None), "nct_to_btc")
        check(lib.ns2vc_k_btc_to_nct(d_y.ptr, cpad, C_, T, B, d_z.ptr, None), "btc_to_nct")
        sync()
        y = d_y.to_numpy((B, T, cpad))
        assert np.array_equal(y[:, :, :C_], x.transpose(0, 2, 1))
        assert np.all(y[:, :, C_:] == 0)
        assert np.array_equal(d_z.to_numpy((B, C_, T)), x)
