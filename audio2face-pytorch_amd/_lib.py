"""ctypes binding of liba2f_hip.so (C ABI declared in include/a2f_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.  Every compute entry of
the product goes through here; the CPU oracle under ``oracle/`` is never imported by the package.
"""
from __future__ import annotations

import ctypes as C
import os

from ._build import LIB_PATH

c_f32p = C.c_void_p          # device pointers travel as integers (tensor.data_ptr())
i64 = C.c_int64
i32 = C.c_int32
f32 = C.c_float
sz = C.c_size_t


class GemmDesc(C.Structure):
    """Mirror of ``a2f_gemm_desc`` (include/a2f_hip.h)."""
    _fields_ = [
        ("A", c_f32p), ("a_off", i64), ("a_rs", i64), ("a_cs", i64),
        ("B", c_f32p), ("b_off", i64), ("b_rs", i64), ("b_cs", i64),
        ("C", c_f32p), ("ldc", i64),
        ("M", i64), ("N", i64), ("K", i64),
        ("bias", c_f32p),
        ("aux", c_f32p), ("ldaux", i64),
        ("ones_out", c_f32p),
        ("act", i32), ("aux_mode", i32), ("accumulate", i32), ("split_k", i32),
        ("mask_operand", i32), ("mask_period", i32), ("mask_phase", i32), ("mask_lo", i32), ("mask_hi", i32),
        ("c_unpack_kh", i32),
        ("splitk_ws", c_f32p),
        ("precision", i32), ("reserved", i32),
    ]


class MMOperand(C.Structure):
    """Mirror of ``a2f_mm_operand`` (include/a2f_hip.h)."""
    _fields_ = [("p", c_f32p), ("bytes", i64), ("ld", i64), ("trans", i32), ("format", i32),
                ("lines_per_seq", i64), ("seq_stride", i64), ("off", i64), ("seq_elems", i64), ("p_lo", c_f32p)]


class MMDesc(C.Structure):
    """Mirror of ``a2f_mm_desc`` (include/a2f_hip.h)."""
    _fields_ = [
        ("A", MMOperand), ("B", MMOperand),
        ("C", c_f32p), ("ldc", i64), ("c_rows_per_seq", i64), ("c_seq_stride", i64),
        ("C2", c_f32p),
        ("M", i64), ("N", i64), ("K", i64),
        ("bias", c_f32p),
        ("aux", c_f32p), ("ldaux", i64), ("aux_bstride", i64),
        ("act", i32), ("aux_mode", i32), ("precision", i32), ("split_k", i32),
        ("ws", c_f32p),
        ("batch", i32), ("accumulate", i32),
        ("a_bstride", i64), ("b_bstride", i64), ("c_bstride", i64), ("bias_bstride", i64),
        ("alpha", f32), ("c_format", i32),
        ("keep_bits", c_f32p), ("keep_wpr", i64),
        ("keep_scale", f32), ("reserved2", i32),
        ("a_rowsum", c_f32p),
        ("c_colsum", c_f32p),
    ]


class RelayoutItem(C.Structure):
    """Mirror of ``a2f_relayout_item`` (include/a2f_hip.h)."""
    _fields_ = [("src", c_f32p), ("dst", c_f32p), ("Co", i32), ("Ci", i32), ("KH", i32), ("Cop", i32), ("Cip", i32), ("mode", i32),
                ("pad_value", f32), ("reserved", i32)]


# name -> (restype, argtypes); must list every symbol include/a2f_hip.h declares
SIGNATURES = {
    "a2f_version": (C.c_int, []),
    "a2f_last_error": (C.c_char_p, []),
    "a2f_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "a2f_wgrad_supported": (C.c_int, [i64, i64]),
    "a2f_wgrad_ws_floats": (sz, [i64, i64, i64]),
    "a2f_wgrad_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "a2f_wgrad_batch_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_int32, C.c_void_p]),
    "a2f_vocaloss_ws_bytes": (sz, [i64, i64]),
    "a2f_vocaloss_fwd": (C.c_int, [c_f32p, c_f32p, i64, i64, f32, f32, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_vocaloss_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, i64, i64, f32, f32, c_f32p, C.c_void_p]),
    "a2f_head_fused_ws_bytes": (sz, [i64, i64, i64]),
    "a2f_head_fused": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, c_f32p, i64, i64, i64, i64, f32, f32,
                                 c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_profile_enable": (C.c_int, [i32]),
    "a2f_profile_head_ms": (C.c_int, [C.POINTER(C.c_float), C.POINTER(i32)]),
    "a2f_voca_pack": (C.c_int, [c_f32p, c_f32p, i64, i64, c_f32p, c_f32p, C.c_void_p]),
    "a2f_voca_trunk_fwd": (C.c_int, [c_f32p, c_f32p, i64, i64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "a2f_voca_trunk_fwd_seq": (C.c_int, [c_f32p, i64, i64, c_f32p, i64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "a2f_interp_linear_fwd": (C.c_int, [c_f32p, i64, i64, i64, i64, c_f32p, C.c_void_p]),
    "a2f_interp_linear_bwd": (C.c_int, [c_f32p, i64, i64, i64, i64, c_f32p, C.c_void_p]),
    "a2f_voca_trunk_bwd_data": (C.c_int, [c_f32p] * 12 + [i64] + [c_f32p] * 6 + [C.c_void_p]),
    "a2f_a2m_pack": (C.c_int, [c_f32p, c_f32p, i64, i64, c_f32p, c_f32p, i64, i64, C.c_void_p]),
    "a2f_conv_weight_pack": (C.c_int, [c_f32p, c_f32p, i64, i64, i64, i32, C.c_void_p]),
    "a2f_relayout_batch": (C.c_int, [C.POINTER(RelayoutItem), i32, C.c_void_p]),
    "a2f_bn_ws_bytes": (sz, [i64, i64]),
    "a2f_bn_fwd": (C.c_int, [c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, c_f32p, f32, f32, i32, i32,
                             c_f32p, c_f32p, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_bn_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, i32, i32,
                             c_f32p, c_f32p, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_bn_fwd_conv0": (C.c_int, [c_f32p, c_f32p, c_f32p, i64, i64, i64, c_f32p, c_f32p, c_f32p, c_f32p, f32, f32, i32, i32,
                                   c_f32p, c_f32p, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_bn_bwd_conv0": (C.c_int, [c_f32p, c_f32p, c_f32p, i64, c_f32p, c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, i32, i32,
                                   c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_ff_save_bytes": (sz, [i64, i64]),
    "a2f_ff_decoder_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i64, i32, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_ff_decoder_bwd": (C.c_int, [c_f32p, c_f32p, i64, i64, i32, C.c_void_p, sz, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "a2f_ff_decoder_fwd_drop": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i64, i32, c_f32p, C.c_void_p, sz, c_f32p, c_f32p, f32, C.c_void_p]),
    "a2f_ff_decoder_bwd_drop": (C.c_int, [c_f32p, c_f32p, i64, i64, i32, C.c_void_p, sz, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, f32, C.c_void_p]),
    "a2f_adam_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, f32, f32, f32, f32, f32, f32, i32, C.c_void_p]),
    "a2f_mm": (C.c_int, [C.POINTER(MMDesc), C.c_void_p]),
    "a2f_split_img": (C.c_int, [c_f32p, i64, i64, i64, i32, C.c_void_p, i64, C.c_void_p]),
    "a2f_posconv_nt": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i64, i64, i64, i64, i64, i32, C.c_void_p]),
    "a2f_posconv_wgrad": (C.c_int, [c_f32p, c_f32p, c_f32p, i64, i64, i64, i64, i64, i64, C.c_void_p]),
    "a2f_layernorm_fwd": (C.c_int, [c_f32p, i64, i64, c_f32p, c_f32p, f32, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "a2f_layernorm_fwd_img": (C.c_int, [c_f32p, i64, i64, c_f32p, c_f32p, f32, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "a2f_layernorm_bwd_img": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, f32, c_f32p,
                                        C.c_void_p, sz, C.c_void_p]),
    "a2f_colsum_img": (C.c_int, [c_f32p, i64, i64, i64, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_attn_fwd_img": (C.c_int, [c_f32p, i32, i64, i64, i64, i64, f32, i32, c_f32p, i64, c_f32p, c_f32p, f32, c_f32p, C.c_void_p]),
    "a2f_attn_bwd_img": (C.c_int, [c_f32p, c_f32p, c_f32p, i32, c_f32p, i64, i64, i64, i64, i64, f32, i32, c_f32p, c_f32p, c_f32p, c_f32p, f32, i32,
                                   C.c_void_p]),
    "a2f_layernorm_bwd_ws_bytes": (sz, [i64, i64]),
    "a2f_layernorm_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_colsum_ws_bytes": (sz, [i64, i64]),
    "a2f_colsum": (C.c_int, [c_f32p, i64, i64, i64, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_gelu_bwd": (C.c_int, [c_f32p, c_f32p, i64, c_f32p, C.c_void_p]),
    "a2f_attn_fwd": (C.c_int, [c_f32p, i64, i64, i64, i64, f32, i32, c_f32p, i64, c_f32p, C.c_void_p]),
    "a2f_attn_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i64, i64, i64, i64, f32, i32, c_f32p, c_f32p, C.c_void_p]),
    "a2f_attn_fwd_drop": (C.c_int, [c_f32p, i64, i64, i64, i64, f32, i32, c_f32p, i64, c_f32p, c_f32p, f32, C.c_void_p]),
    "a2f_attn_bwd_drop": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i64, i64, i64, i64, f32, i32, c_f32p, c_f32p, c_f32p, c_f32p, f32,
                                    C.c_void_p]),
    "a2f_keep_bits": (C.c_int, [C.c_uint64, C.c_uint32, i64, i64, f32, c_f32p, i64, C.c_void_p]),
    "a2f_keep_bits_dev": (C.c_int, [c_f32p, C.c_uint32, i64, i64, f32, c_f32p, i64, C.c_void_p]),
    "a2f_keep_bits_attn_dev": (C.c_int, [c_f32p, C.c_uint32, i64, i64, f32, c_f32p, c_f32p, C.c_void_p]),
    "a2f_attn_keep_words": (i64, [i64]),
    "a2f_keep_bits_attn": (C.c_int, [C.c_uint64, C.c_uint32, i64, i64, f32, c_f32p, c_f32p, C.c_void_p]),
    "a2f_dropout_apply": (C.c_int, [c_f32p, c_f32p, i64, i64, i64, i64, f32, c_f32p, C.c_void_p]),
    "a2f_featenc0_ws_bytes": (sz, [i64, i64, i64]),
    "a2f_featenc0_fwd": (C.c_int, [c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, i64, f32, c_f32p, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_featenc0_fwd_img": (C.c_int, [c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, i64, f32, c_f32p, i32, c_f32p, C.c_void_p, sz, C.c_void_p]),
    "a2f_featenc0_bwd": (C.c_int, [c_f32p, i64, i64, c_f32p, c_f32p, c_f32p, i64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, sz,
                                   C.c_void_p]),
    "a2f_stream_copy": (C.c_int, [c_f32p, c_f32p, i64, C.c_void_p]),
    "a2f_stream_triad": (C.c_int, [c_f32p, c_f32p, c_f32p, i64, C.c_void_p]),
}

_LIB = None


class A2FError(RuntimeError):
    pass


def load(path: str | None = None):
    """Load (once) and return the ctypes handle; raises A2FError if the library is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = path or os.environ.get("A2F_LIB_PATH") or LIB_PATH        # A2F_LIB_PATH: tooling hook for ablation builds
    if not os.path.exists(path):
        raise A2FError(f"HIP extension not built: {path} is missing. Run `python __graft_entry__.py` (build()) "
                       f"or `python audio2face-pytorch_amd/_build.py`. There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().a2f_last_error()
        raise A2FError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
