"""ctypes loader for libcrab_hip.so (the C-ABI in include/crab_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the HIP library is missing or an entry point
fails, a CrabHipError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRAB_HIP_LIB") or os.path.join(_HERE, "libcrab_hip.so")      # override: A/B builds of the same C-ABI


class CrabHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
        ("A2", C.c_void_p), ("B2", C.c_void_p),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64), ("lda2", C.c_int64),
        ("ldb2", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("K2", C.c_int32),
        ("act", C.c_int32), ("c_fp32", C.c_int32), ("res_scale", C.c_float),
        ("batch", C.c_int32), ("nb0", C.c_int32),
        ("sA0", C.c_int64), ("sA1", C.c_int64), ("sB0", C.c_int64), ("sB1", C.c_int64), ("sC0", C.c_int64),
        ("sC1", C.c_int64), ("sR0", C.c_int64), ("sR1", C.c_int64), ("sBias0", C.c_int64), ("sBias1", C.c_int64),
        ("tune", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("norm_w", C.c_void_p), ("norm_out", C.c_void_p), ("ld_norm", C.c_int64), ("norm_eps", C.c_float),
        ("rope_tab", C.c_void_p), ("rope_k_cache", C.c_void_p), ("rope_v_cache", C.c_void_p), ("rope_pos_dev", C.c_void_p),
        ("rope_H", C.c_int32), ("rope_Hk", C.c_int32), ("rope_d", C.c_int32), ("rope_Tmax", C.c_int32), ("rope_pos0", C.c_int32),
        ("route_RA", C.c_void_p), ("route_U", C.c_void_p), ("route_ldra", C.c_int64), ("route_ldu", C.c_int64),
        ("route_nproj", C.c_int32), ("route_nl", C.c_int32), ("route_r", C.c_int32), ("route_ucols", C.c_int32), ("route_scaling", C.c_float),
        ("lora_RA", C.c_void_p), ("lora_ldra", C.c_int64), ("lora_nl", C.c_int32), ("lora_r", C.c_int32), ("lora_scaling", C.c_float),
        ("rope_S", C.c_int32), ("rope_ld_pos", C.c_int64), ("rope_pos_ids", C.c_void_p),
        ("rope_vt", C.c_void_p), ("rope_vt_ld", C.c_int64),
        ("r_fp32", C.c_int32),
        ("rope_row_off", C.c_void_p),
        ("norm_w_fp32", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("o", C.c_void_p),
        ("q_bs", C.c_int64), ("q_hs", C.c_int64), ("q_ss", C.c_int64),
        ("k_bs", C.c_int64), ("k_hs", C.c_int64), ("k_ss", C.c_int64),
        ("vt_bs", C.c_int64), ("vt_hs", C.c_int64), ("vt_ds", C.c_int64),
        ("o_bs", C.c_int64), ("o_ss", C.c_int64),
        ("bias", C.c_void_p), ("gate", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("Hk", C.c_int32), ("Sq", C.c_int32), ("Skv", C.c_int32),
        ("head_dim", C.c_int32), ("causal", C.c_int32), ("scale", C.c_float),
        ("kv_start", C.c_void_p),
        ("key_mask", C.c_void_p), ("key_mask_ld", C.c_int64),
    ]


class Dense(C.Structure):
    """crab_dense"""
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("ldw", C.c_int64), ("N", C.c_int32), ("K", C.c_int32)]


class LN(C.Structure):
    """crab_ln"""
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("eps", C.c_float), ("fp32", C.c_int32)]


class ClipLayerW(C.Structure):
    _fields_ = [("ln1", LN), ("ln2", LN), ("qkv", Dense), ("out", Dense), ("fc1", Dense), ("fc2", Dense), ("H", C.c_int32)]


class BeatsLayerW(C.Structure):
    _fields_ = [("qkv", Dense), ("out", Dense), ("fc1", Dense), ("fc2", Dense), ("ln_attn", LN), ("ln_final", LN),
                ("grep_w", C.c_void_p), ("grep_b", C.c_void_p), ("grep_a", C.c_void_p), ("H", C.c_int32), ("alpha", C.c_float)]


class QformerLayerW(C.Structure):
    _fields_ = [("sq", Dense), ("skv", Dense), ("so", Dense), ("sln", LN), ("cq", Dense), ("ckv", Dense), ("co", Dense), ("cln", LN),
                ("iq", Dense), ("oq", Dense), ("oln", LN), ("H", C.c_int32)]


class EncIO(C.Structure):
    """crab_enc_io"""
    _fields_ = [("x", C.c_void_p), ("a", C.c_void_p), ("y", C.c_void_p), ("qkv", C.c_void_p), ("att", C.c_void_p), ("f", C.c_void_p),
                ("vt", C.c_void_p), ("vt_bytes", C.c_int64), ("enc", C.c_void_p), ("enc_rows", C.c_int32),
                ("bias", C.c_void_p), ("gate", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("B", C.c_int32), ("S", C.c_int32), ("x_fp32", C.c_int32)]


class LinearGroup(C.Structure):
    """crab_linear_group"""
    _fields_ = [
        ("W", C.c_void_p), ("bias", C.c_void_p), ("RA", C.c_void_p), ("B2", C.c_void_p),
        ("ldw", C.c_int64), ("ldra", C.c_int64), ("ldb2", C.c_int64),
        ("N", C.c_int32), ("K", C.c_int32), ("nproj", C.c_int32), ("nl", C.c_int32), ("r", C.c_int32), ("tcols", C.c_int32),
        ("ucols", C.c_int32), ("scaling", C.c_float),
    ]


class LlamaLayer(C.Structure):
    """crab_llama_layer"""
    _fields_ = [
        ("qkv", LinearGroup), ("o", LinearGroup), ("gu", LinearGroup), ("down", LinearGroup),
        ("post_attention_norm_w", C.c_void_p), ("next_norm_w", C.c_void_p), ("next_qkv", C.POINTER(LinearGroup)),
        ("H", C.c_int32), ("Hk", C.c_int32), ("d", C.c_int32), ("rms_eps", C.c_float),
        ("norm_w_fp32", C.c_int32),
    ]


class LlamaIO(C.Structure):
    """crab_llama_io"""
    _fields_ = [
        ("x", C.c_void_p), ("h", C.c_void_p), ("qkv", C.c_void_p), ("att", C.c_void_p), ("act", C.c_void_p), ("u", C.c_void_p),
        ("u2", C.c_void_p),
        ("ldx", C.c_int64), ("ldh", C.c_int64), ("ldqkv", C.c_int64), ("ldatt", C.c_int64), ("ldact", C.c_int64), ("ldu", C.c_int64),
        ("route_ws", C.c_void_p), ("route_ws_bytes", C.c_int64), ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("rope_tab", C.c_void_p), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p), ("cache_layer_stride", C.c_int64),
        ("vt", C.c_void_p), ("vt_ld", C.c_int64), ("pos_dev", C.c_void_p),
        ("B", C.c_int32), ("S", C.c_int32), ("Tmax", C.c_int32), ("pos0", C.c_int32), ("u_qkv_ready", C.c_int32),
        ("attn_ws", C.c_void_p), ("attn_ws_bytes", C.c_int64),
        ("x_fp32", C.c_int32),
        ("row_off", C.c_void_p),
        ("pos_ids", C.c_void_p), ("ld_pos", C.c_int64), ("kv_start", C.c_void_p),
        ("last_rows_only", C.c_int32),
    ]


# every symbol include/crab_hip.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "crab_abi_version": (_i, []),
    "crab_sizeof_gemm_desc": (_i, []),
    "crab_decode_max_rows": (_i, []),
    "crab_attn_split_below": (_i, []),
    "crab_gemm_fuses_prefill_rope": (_i, [_vp]),
    "crab_sizeof_attn_desc": (_i, []),
    "crab_sizeof_llama_layer": (_i, []),
    "crab_sizeof_llama_io": (_i, []),
    "crab_clip_layer": (_i, [_vp, _vp, C.POINTER(ClipLayerW), C.POINTER(EncIO)]),
    "crab_beats_layer": (_i, [_vp, _vp, C.POINTER(BeatsLayerW), C.POINTER(EncIO)]),
    "crab_qformer_layer": (_i, [_vp, _vp, C.POINTER(QformerLayerW), C.POINTER(EncIO)]),
    "crab_sizeof_enc_io": (_i, []),
    "crab_sizeof_clip_layer_w": (_i, []),
    "crab_sizeof_beats_layer_w": (_i, []),
    "crab_sizeof_qformer_layer_w": (_i, []),
    "crab_llama_layer_prefill": (_i, [_vp, _vp, C.POINTER(LlamaLayer), C.POINTER(LlamaIO), _i]),
    "crab_llama_layer_decode": (_i, [_vp, _vp, C.POINTER(LlamaLayer), C.POINTER(LlamaIO), _i]),
    "crab_llama_layers": (_i, [_vp, _vp, C.POINTER(LlamaLayer), _i, C.POINTER(LlamaIO)]),
    "crab_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "crab_ctx_destroy": (None, [_vp]),
    "crab_last_error": (C.c_char_p, [_vp]),
    "crab_sync": (_i, [_vp, _vp]),
    "crab_trace_begin": (_i, [_vp]),
    "crab_trace_end": (_i64, [_vp, C.c_char_p, _i64]),
    "crab_gemm_bf16": (_i, [_vp, _vp, C.POINTER(GemmDesc)]),
    "crab_rowfin_workspace": (_i64, [_i, _i]),
    "crab_rowfin_lora_ok": (_i, [_i, _i, _i]),
    "crab_attn_decode_rope_workspace": (_i64, [_i, _i, _i]),
    "crab_attn_decode_rope": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _i64]),
    "crab_hyperlora_mix": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _i, _f]),
    "crab_hyperlora_route_workspace": (_i64, [_i, _i, _i]),
    "crab_hyperlora_route": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _f, _vp, _i64]),
    "crab_rmsnorm": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _f]),
    "crab_layernorm": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _f]),
    "crab_embedding": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i]),
    "crab_embedding_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i]),
    "crab_rmsnorm_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _f]),
    "crab_layernorm_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _f]),
    "crab_rmsnorm_p": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _i, _vp, _i64, _i, _i, _f]),
    "crab_layernorm_p": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _i, _vp, _i64, _i, _i, _f]),
    "crab_clip_embed_ln_p": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _f]),
    "crab_cast_rows_bf16_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i]),
    "crab_cast_rows_f32_bf16": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i]),
    "crab_rope_table": (_i, [_vp, _vp, _vp, _i, _i, _f]),
    "crab_qkv_rope_split": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "crab_qkv_rope_split_ids": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64]),
    "crab_qkv_rope_split_ragged": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "crab_attn_decode_masked": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    "crab_attn_decode_keymask": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _i64]),
    "crab_attn_fwd": (_i, [_vp, _vp, C.POINTER(AttnDesc)]),
    "crab_attn_decode": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _f]),
    "crab_swiglu": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i]),
    "crab_argmax": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _i]),
    "crab_im2col_patch": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i, _i, _i]),
    "crab_clip_embed_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f]),
    "crab_beats_posconv_pad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "crab_beats_relpos_bias": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "crab_beats_gru_gate": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "crab_copy_rows": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i]),
    "crab_cast_f32_bf16": (_i, [_vp, _vp, _vp, _vp, _i64]),
    "crab_copy_rows_batched": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i]),
    "crab_greedy_select": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _i]),
    "crab_advance": (_i, [_vp, _vp, _vp, _vp]),
    "crab_sample_select": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _f, _i, _f, C.c_uint64]),
    "crab_im2col3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "crab_pixel_shuffle2x": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    "crab_bilinear": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _i64, _i, _i, _i, _vp, _i, _i, _f, _f]),
    "crab_dense_pe": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i]),
    "crab_add_rows": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _vp, _i64, _i, _i]),
    "crab_mask_gate": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _i64, _i, _i]),
    "crab_act_inplace": (_i, [_vp, _vp, _vp, _i64, _i]),
    "crab_dist_unique_id": (_i, [_vp, _vp]),
    "crab_dist_init": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "crab_gather_results": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _i]),
    "crab_dist_world": (_i, [_vp]),
    "crab_dist_rank": (_i, [_vp]),
    "crab_dist_destroy": (None, [_vp]),
    "crab_mask_labels": (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    "crab_group_mean": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _f]),
    "crab_mask_iou": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _f, _vp, _vp]),
    "crab_fmeasure": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp, _i, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "crab_miou_fscore": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crab_color_to_label": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _vp]),
    "crab_bicubic_ksize": (_i, [_i, _i]),
    "crab_bicubic_coeffs": (_i, [_i, _i, _vp, _vp, _i]),
    "crab_resample_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i]),
    "crab_clip_normalize": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _f]),
    "crab_im2col3x3_strided": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "crab_groupnorm_workspace": (_i64, [_i, _i, _i]),
    "crab_groupnorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _i64]),
    "crab_upsample_nearest2x": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "crab_softmax_rows": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _f]),
    "crab_row_sqnorm": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "crab_vq_argmin": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp, _i64]),
    "crab_row_sqnorm_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "crab_vq_nearest_f32_workspace": (_i64, [_i, _i]),
    "crab_vq_nearest_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _vp, _i64, _vp, _i64]),
    "crab_groupnorm_p": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _i, _i, _vp, _i64]),
    "crab_split3": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i]),
    "crab_groupnorm_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _i64]),
    "crab_add_bias_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i]),
    "crab_softmax_rows_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _f]),
    "crab_kaldi_fbank_frames": (_i, [_i]),
    "crab_kaldi_fbank": (_i, [_vp, _vp, _vp, _i64, _i, _i, _f, _f, _vp, _vp, _vp, _f, _f]),
}

def header_const(name: str) -> int:
    """An integer `#define` of include/crab_hip.h: the single source of the dispatch bounds the Python side mirrors (CRAB_DECODE_MAX_ROWS,
    CRAB_ATTN_SPLIT_BELOW); load() checks them against the values compiled into the library."""
    import re
    with open(os.path.join(os.path.dirname(_HERE), "include", "crab_hip.h")) as f:
        m = re.search(r"^#define\s+" + re.escape(name) + r"\s+(\d+)", f.read(), re.M)
    if not m:
        raise CrabHipError(f"include/crab_hip.h does not define {name}")
    return int(m.group(1))


_lib = None
_lock = threading.Lock()
_ctxs = {}


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every declared symbol.  Raises CrabHipError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise CrabHipError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(crab_amd/csrc/build.sh). There is no fallback path.")
        # torch FIRST: libcrab_hip.so needs libamdhip64, and PyTorch-ROCm bundles its own.  Whichever is loaded first serves the whole process (same
        # SONAME); with the library loaded before torch (r06: build() followed by smoke() in one process) the library talked to /opt/rocm's runtime and
        # torch to its own - crab_ctx_create then saw no device (hipGetDeviceCount of the other runtime) while torch.cuda.is_available() was True
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name, None)
            if fn is None:
                raise CrabHipError(f"libcrab_hip.so does not export {name}")
            fn.restype = res
            fn.argtypes = args
        for macro, fn in (("CRAB_DECODE_MAX_ROWS", lib.crab_decode_max_rows), ("CRAB_ATTN_SPLIT_BELOW", lib.crab_attn_split_below)):
            if fn() != header_const(macro):
                raise CrabHipError(f"libcrab_hip.so was built with {macro} = {fn()}, include/crab_hip.h says {header_const(macro)}: rebuild (csrc/build.sh)")
        _lib = lib
    return _lib


def ctx(device: int = 0) -> int:
    """One crab_ctx per device per process."""
    lib = load()
    if device not in _ctxs:
        h = C.c_void_p()
        rc = lib.crab_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise CrabHipError(f"crab_ctx_create(device={device}) failed with {rc} (no MI355X visible?)")
        _ctxs[device] = h
    return _ctxs[device]


def check(rc: int, device: int = 0):
    if rc != 0:
        msg = load().crab_last_error(_ctxs.get(device))
        raise CrabHipError(f"libcrab_hip error {rc}: {msg.decode() if msg else '?'}")
