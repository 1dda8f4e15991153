"""ctypes binding of libns2vc_hip.so (include/ns2vc_hip.h).

The library is built in-tree (``ns2vc_amd/lib/libns2vc_hip.so``) by
``__graft_entry__.build()`` / ``make -C ns2vc_amd/csrc``.  There is NO fallback:
if the shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libns2vc_hip.so")
NCOEF = 12
MAX_LEVELS = 8
PREC_F32, PREC_BF16, PREC_F16 = 0, 1, 2
ABI_VERSION = 7


class Ns2vcError(RuntimeError):
    pass


class UnetCfg(C.Structure):
    _fields_ = [
        ("latent_channels", C.c_int32), ("content_channels", C.c_int32), ("n_levels", C.c_int32),
        ("block_out_channels", C.c_int32 * MAX_LEVELS), ("norm_num_groups", C.c_int32),
        ("cross_attention_dim", C.c_int32), ("heads", C.c_int32), ("layers_per_block", C.c_int32),
        ("pool_heads", C.c_int32),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p),
        ("lda0", C.c_int32), ("lda1", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
        ("B", C.c_int32), ("Tin", C.c_int32), ("Tout", C.c_int32), ("M", C.c_int32),
        ("taps", C.c_int32), ("tmode", C.c_int32),
        ("w", C.c_void_p), ("K", C.c_int32), ("N", C.c_int32),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ldres", C.c_int32), ("geglu", C.c_int32),
        ("out_f32", C.c_void_p), ("ldo_f32", C.c_int32),
        ("out_op", C.c_void_p), ("ldo_op", C.c_int32),
        ("stats", C.c_void_p),
        ("a2", C.c_void_p), ("lda2", C.c_int32), ("c2", C.c_int32),
        ("rowstats", C.c_void_p),
        ("ln_stats", C.c_void_p), ("ln_wsum", C.c_void_p), ("ln_eps", C.c_float), ("ln_dim", C.c_int32),
        ("ln_health", C.c_void_p),
        ("gnp_x", C.c_void_p), ("gnp_ldx", C.c_int32),
        ("gnp_stats", C.c_void_p), ("gnp_gamma", C.c_void_p), ("gnp_beta", C.c_void_p),
        ("gnp_temb", C.c_void_p), ("gnp_ldtemb", C.c_int32),
        ("gnp_eps", C.c_float), ("gnp_G", C.c_int32), ("gnp_silu", C.c_int32),
        ("gnp_sync", C.c_void_p), ("gnp_alone", C.c_void_p),
        ("gnp_x1", C.c_void_p), ("gnp_ldx1", C.c_int32), ("gnp_c1", C.c_int32), ("gnp_stats1", C.c_void_p), ("gnp_raw", C.c_void_p),
        ("algo", C.c_int32), ("w_tiled", C.c_void_p),
        ("sol_coef", C.c_void_p), ("sol_step", C.c_void_p), ("sol_ncoef", C.c_int32),
        ("sol_xe", C.c_void_p), ("sol_xe_op", C.c_void_p), ("sol_xbar", C.c_void_p), ("sol_d1", C.c_void_p), ("sol_mprev", C.c_void_p), ("sol_ld", C.c_int32),
        ("conv_bn", C.c_int32),
        ("gnp_pair", C.c_int32), ("sol_op_pair", C.c_int32),
    ]


class FfnArgs(C.Structure):
    _fields_ = [
        ("yn", C.c_void_p), ("ldy", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_eps", C.c_float),
        ("wstream", C.c_void_p), ("consts", C.c_void_p), ("bias2", C.c_void_p),
        ("res", C.c_void_p), ("ldres", C.c_int32),
        ("out_f32", C.c_void_p), ("ldo_f32", C.c_int32),
        ("out_op", C.c_void_p), ("ldo_op", C.c_int32),
        ("stats", C.c_void_p),
        ("B", C.c_int32), ("T", C.c_int32), ("M", C.c_int32), ("dim", C.c_int32),
        ("ln_health", C.c_void_p),
        ("pre_a", C.c_void_p), ("pre_lda", C.c_int32),
        ("pre_bias", C.c_void_p),
        ("pre_res", C.c_void_p), ("pre_ldres", C.c_int32),
        ("att_q", C.c_void_p), ("att_ldq", C.c_int32),
        ("att_kv", C.c_void_p), ("att_bias", C.c_void_p), ("att_scale", C.c_float), ("att_Lk", C.c_int32),
    ]


class GegluArgs(C.Structure):
    _fields_ = [
        ("yn", C.c_void_p), ("ldy", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_eps", C.c_float),
        ("wstream", C.c_void_p), ("consts", C.c_void_p),
        ("out_op", C.c_void_p), ("ldo", C.c_int32),
        ("M", C.c_int32), ("dim", C.c_int32),
        ("ln_health", C.c_void_p),
    ]


class RowchainArgs(C.Structure):
    _fields_ = [
        ("a_op", C.c_void_p), ("lda", C.c_int32),
        ("wstream", C.c_void_p), ("bias1", C.c_void_p), ("consts2", C.c_void_p),
        ("res", C.c_void_p), ("ldres", C.c_int32),
        ("out1_f32", C.c_void_p), ("ldo1", C.c_int32),
        ("out2_op", C.c_void_p), ("ldo2", C.c_int32),
        ("ln_eps", C.c_float), ("M", C.c_int32), ("dim", C.c_int32), ("n2", C.c_int32),
        ("ln_health", C.c_void_p),
        ("gn_x", C.c_void_p), ("ldx", C.c_int32),
        ("gn_stats", C.c_void_p), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p),
        ("gn_eps", C.c_float), ("T", C.c_int32), ("G", C.c_int32),
        ("slices", C.c_int32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
        ("bias", C.c_void_p), ("scale", C.c_float),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("pv_fp8", C.c_int32), ("exact_only", C.c_int32), ("fallbacks", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol the header declares
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_I = C.c_int
PROTOTYPES = {
    "ns2vc_abi_version": (_I, []),
    "ns2vc_last_error": (C.c_char_p, []),
    "ns2vc_device_count": (_I, [C.POINTER(_I)]),
    "ns2vc_set_device": (_I, [_I]),
    "ns2vc_device_xcd_round_robin": (_I, [C.POINTER(_I)]),
    "ns2vc_device_name": (_I, [C.c_char_p, _I]),
    "ns2vc_unet_create": (_I, [C.POINTER(UnetCfg), _PP]),
    "ns2vc_unet_destroy": (_I, [_P]),
    "ns2vc_unet_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "ns2vc_unet_finalize_weights": (_I, [_P, _I]),
    "ns2vc_unet_num_missing_weights": (_I, [_P, C.c_char_p, _I]),
    "ns2vc_unet_prepare": (_I, [_P, _I, _I, _I]),
    "ns2vc_unet_workspace_bytes": (_I, [_P, C.POINTER(C.c_size_t)]),
    "ns2vc_unet_set_condition": (_I, [_P, _P, _P, _P, _P]),
    "ns2vc_unet_set_content": (_I, [_P, _P, _P]),
    "ns2vc_unet_set_prompt": (_I, [_P, _P, _P, _P]),
    "ns2vc_unet_set_mask": (_I, [_P, _P, _P]),
    "ns2vc_unet_forward": (_I, [_P, _P, _P, _P, _P]),
    "ns2vc_sampler_load": (_I, [_P, _I, C.POINTER(C.c_float)]),
    "ns2vc_sampler_run": (_I, [_P, _P, _I, _P]),
    "ns2vc_sampler_begin": (_I, [_P, _P, _P]),
    "ns2vc_sampler_steps": (_I, [_P, _I, _I, _P]),
    "ns2vc_sampler_end": (_I, [_P, _P, _P]),
    "ns2vc_sampler_handoff": (_I, [_P, _P, _P]),
    "ns2vc_sampler_peek": (_I, [_P, _P, _P]),
    "ns2vc_unet_attn_fallbacks": (_I, [_P, C.POINTER(C.c_ulonglong), _I, _P]),
    "ns2vc_unet_gn_coop_alone": (_I, [_P, C.POINTER(C.c_ulonglong), _I, _P]),
    "ns2vc_unet_set_debug": (_I, [_P, _I]),
    "ns2vc_unet_set_option": (_I, [_P, C.c_char_p, _I]),
    "ns2vc_unet_ln_ratio": (_I, [_P, C.POINTER(C.c_float), _P]),
    "ns2vc_unet_ln_ratio_post": (_I, [_P, _P]),
    "ns2vc_unet_ln_ratio_poll": (_I, [_P, C.POINTER(C.c_float), C.POINTER(_I)]),
    "ns2vc_unet_num_taps": (_I, [_P]),
    "ns2vc_unet_tap_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_I), C.POINTER(_I)]),
    "ns2vc_unet_tap_read": (_I, [_P, _I, _P]),
    "ns2vc_unet_num_launches": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "ns2vc_unet_op_info": (_I, [_P, _I, _I, C.c_char_p, _I, C.POINTER(_I), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ns2vc_unet_profile_forward": (_I, [_P, C.POINTER(C.c_float), _I, _I, _P]),
    "ns2vc_dev_malloc": (_I, [_PP, C.c_size_t]),
    "ns2vc_dev_free": (_I, [_P]),
    "ns2vc_memcpy_h2d": (_I, [_P, _P, C.c_size_t]),
    "ns2vc_memcpy_d2h": (_I, [_P, _P, C.c_size_t]),
    "ns2vc_dev_sync": (_I, []),
    "ns2vc_stream_create": (_I, [_PP]),
    "ns2vc_stream_create_cu_mask": (_I, [_PP, C.POINTER(C.c_uint32), _I]),
    "ns2vc_stream_destroy": (_I, [_P]),
    "ns2vc_stream_sync": (_I, [_P]),
    "ns2vc_event_create": (_I, [_PP]),
    "ns2vc_event_destroy": (_I, [_P]),
    "ns2vc_event_record": (_I, [_P, _P]),
    "ns2vc_event_elapsed_ms": (_I, [_P, _P, C.POINTER(C.c_float)]),
    "ns2vc_pack_weight": (_I, [_P, _I, _I, _I, _PP]),
    "ns2vc_k_gemm": (_I, [C.POINTER(GemmArgs), _I, _P]),
    "ns2vc_pack_conv3_tiled": (_I, [_P, _I, _I, _I, _I, _PP]),
    "ns2vc_weight_rowsum": (_I, [_P, _I, _I, _I, _PP]),
    "ns2vc_debug_set_gemm_trace": (_I, [_P]),
    "ns2vc_debug_set_gemm_tile": (_I, [_I, _I, _I]),
    "ns2vc_debug_placement": (_I, [_P, _I, _I, C.POINTER(C.c_uint32)]),
    "ns2vc_debug_poison": (_I, [C.c_uint, _I, _P]),
    "ns2vc_k_attention": (_I, [C.POINTER(AttnArgs), _I, _I, _P]),
    "ns2vc_pack_ffn": (_I, [_P, _P, _I, _I, _PP]),
    "ns2vc_pack_ffn_pre": (_I, [_P, _P, _P, _I, _I, _PP]),
    "ns2vc_xattn_pack_bytes": (C.c_size_t, [_I, _I, _I]),
    "ns2vc_k_xattn_pack": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P]),
    "ns2vc_k_ffn": (_I, [C.POINTER(FfnArgs), _I, _P]),
    "ns2vc_pack_geglu": (_I, [_P, _P, _I, _I, _PP, _PP]),
    "ns2vc_k_geglu": (_I, [C.POINTER(GegluArgs), _I, _P]),
    "ns2vc_pack_geglu_host": (_I, [_P, _P, _I, _I, _P, _P]),
    "ns2vc_debug_set_geglu_min_rows": (_I, [_I]),
    "ns2vc_pack_rowchain": (_I, [_P, _P, _I, _I, _I, _PP]),
    "ns2vc_pack_rowchain_sliced": (_I, [_P, _P, _I, _I, _I, _I, _PP]),
    "ns2vc_k_rowchain": (_I, [C.POINTER(RowchainArgs), _I, _P]),
    "ns2vc_debug_set_rowchain_tokens": (_I, [_I]),
    "ns2vc_debug_set_attn_keys": (_I, [_I]),
    "ns2vc_debug_set_attn_optimistic": (_I, [_I]),
    "ns2vc_k_groupnorm_stats": (_I, [_P, _I, _I, _P, _I, _I, _I, C.c_float, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "ns2vc_k_groupnorm_stats2": (_I, [_P, _I, _I, _P, _P, _I, _I, _P, _I, _I, _I, C.c_float, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "ns2vc_k_groupnorm": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, C.c_float, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "ns2vc_k_layernorm_apply": (_I, [_P, _I, _I, _I, C.c_float, _P, _I, _P]),
    "ns2vc_to_operand": (_I, [_P, C.c_size_t, _I, _PP]),
    "ns2vc_from_operand": (_I, [_P, C.c_size_t, _I, _P]),
    "ns2vc_round_to_operand": (_I, [_P, C.c_size_t, _I, _P]),
    "ns2vc_k_nct_to_btc": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "ns2vc_k_btc_to_nct": (_I, [_P, _I, _I, _I, _I, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load(path: Optional[str] = None) -> C.CDLL:
    """dlopen the engine and bind every prototype.  Raises if the .so is absent
    (build it with ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("NS2VC_LIB") or LIB_PATH      # NS2VC_LIB: A/B a variant build (make OUT=../lib/variants/x)
    if not os.path.exists(p):
        raise Ns2vcError(f"{p} not found: the HIP engine is not built. Run __graft_entry__.build() "
                         f"(or `make -C ns2vc_amd/csrc`). There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64; if this library pulled in the system one first, a later
    # `import torch` would bring a second HIP runtime into the process and torch would see no devices.  Let torch (when
    # present) load its runtime first -- the engine then binds to the already-loaded HIP symbols.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ns2vc_abi_version() != ABI_VERSION:
        raise Ns2vcError(f"ABI version mismatch: library reports {lib.ns2vc_abi_version()}, binding expects {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ns2vc_last_error()
        raise Ns2vcError(f"{what}: {msg.decode() if msg else 'unknown error'}")
