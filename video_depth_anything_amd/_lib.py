"""ctypes binding of libvda_hip.so (include/vda.h). There is no CPU fallback:
if the library is missing or does not export a symbol, importing this fails."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDA_LIB_PATH: load another build of the library instead (tools/lib_ab.py runs two builds side by side in one process - e.g. the
# previous commit's against the working tree's - with VDA_LIB_TOLERANT=1 so that a symbol only one of them exports is simply absent)
LIB_PATH = os.environ.get("VDA_LIB_PATH") or os.path.join(_HERE, "libvda_hip.so")

A_DENSE, A_CONV3X3 = 0, 1
(EPI_BIAS_F16, EPI_BIAS_GELU_F16, EPI_BIAS_RELU_F16, EPI_SCALE_RES_F32, EPI_RES_F16, EPI_GEGLU_F16,
 EPI_PATCH_F32, EPI_CONVT_F16, EPI_BIAS_F32, EPI_SCALE_RES_F32_H, EPI_SCALE_RES_SPLIT, EPI_LN_BIAS_F16, EPI_LN_GELU_F16) = range(13)


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("res", C.c_void_p), ("res2", C.c_void_p), ("gamma", C.c_void_p), ("pos", C.c_void_p),
        ("zero_page", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldc", C.c_int32),
        ("a_mode", C.c_int32), ("epilogue", C.c_int32), ("relu_in", C.c_int32),
        ("cB", C.c_int32), ("cH", C.c_int32), ("cW", C.c_int32), ("cCin", C.c_int32),
        ("cHo", C.c_int32), ("cWo", C.c_int32), ("cStride", C.c_int32),
        ("P", C.c_int32), ("tK", C.c_int32), ("tH", C.c_int32), ("tW", C.c_int32), ("tCout", C.c_int32),
        ("out2", C.c_void_p), ("stats", C.c_void_p), ("sched", C.c_void_p),
        ("stats_ld", C.c_int32), ("tile_rows", C.c_int32),
    ]


class Config(C.Structure):
    """vda_config (include/vda.h)."""
    _fields_ = [("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32), ("taps", C.c_int32 * 4),
                ("features", C.c_int32), ("out_channels", C.c_int32 * 4), ("num_frames", C.c_int32), ("use_clstoken", C.c_int32),
                ("use_bn", C.c_int32), ("pe_rope", C.c_int32)]


PREC_F16, PREC_F32 = 0, 1
DTYPE_F32 = 0

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
SIGNATURES = {
    "vda_last_error": (C.c_char_p, []),
    "vda_abi_version": (_i, []),
    "vda_gemm_f16": (_i, [C.POINTER(GemmArgs), _vp]),
    "vda_gemm_f32": (_i, [C.POINTER(GemmArgs), _vp]),
    "vda_gemm_set_variant": (_i, [_i]),
    "vda_gemm_set_debug": (_i, [_i]),
    "vda_set_max_wgs": (_i, [_i]),
    "vda_gemm_plan_split": (_i, [_i, _i, _i, _i, _i]),
    "vda_gemm_row_range": (_i, [C.POINTER(GemmArgs), _i, _i, C.POINTER(GemmArgs)]),
    "vda_gemm_last_kernel": (C.c_char_p, []),
    "vda_layernorm_f32_f16": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "vda_layernorm_residual_f32_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp]),
    "vda_layernorm_f32_f32": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "vda_split_stats_f32": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "vda_split_center_stats_f32": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "vda_ln_stats_finalize": (_i, [_vp, _vp, _f, _i, _i, _vp, _vp]),
    "vda_layernorm_split_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp]),
    "vda_fold_ln_weight": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "vda_mlp_fused_supported": (_i, [_i, _i]),
    "vda_mlp_permute_w2_f16": (_i, [_vp, _vp, _i, _i, _vp]),
    "vda_mlp_fused_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vda_groupnorm_nhwc_f16": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _i, _vp]),
    "vda_groupnorm_nhwc_f32": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _i, _vp]),
    "vda_attention_f16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vda_attention_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vda_attention_set_variant": (_i, [_i]),
    "vda_depth_tail_set_variant": (_i, [_i]),
    "vda_conv3x3_up2_set_variant": (_i, [_i]),
    "vda_conv_lds_set_variant": (_i, [_i]),
    "vda_temporal_attention_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vda_temporal_attention_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vda_temporal_attention_set_variant": (_i, [_i]),
    "vda_rope_qk_f16": (_i, [_vp, _i, _i, _i, _vp]),
    "vda_rope_qk_f32": (_i, [_vp, _i, _i, _i, _vp]),
    "vda_bilinear_nhwc_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vda_bilinear_nhwc_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vda_bilinear_plane_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vda_patchify_f32_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vda_patchify_f32_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vda_pos_embed_resample_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vda_cls_rows_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "vda_readout_concat_f16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vda_readout_concat_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vda_head_out_f16_f32": (_i, [_vp, _vp, _f, _vp, _i, _i, _vp]),
    "vda_head_out_f32_f32": (_i, [_vp, _vp, _f, _vp, _ll, _i, _vp]),
    "vda_conv3x3_up2_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vda_depth_tail_f16": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vda_normalize_u8_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vda_gather_normalize_u8_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vda_gather_resize_normalize_u8_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vda_lsq_scale_shift_f32": (_i, [_vp, _vp, C.c_longlong, _vp, _i, _vp, _vp]),
    "vda_stitch_window_f32": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp, _vp]),
    "vda_affine_clamp_f32": (_i, [_vp, _vp, _vp, C.c_longlong, _vp]),
    # handle API
    "vda_create": (_i, [C.POINTER(Config), C.POINTER(_vp)]),
    "vda_destroy": (_i, [_vp]),
    "vda_num_weights": (_i, [_vp]),
    "vda_load_weight": (_i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i, _i]),
    "vda_finalize_weights": (_i, [_vp]),
    "vda_workspace_bytes": (C.c_int64, [_vp, _i, _i, _i, _i, _i]),
    "vda_set_workspace": (_i, [_vp, _vp, C.c_int64]),
    "vda_prepare": (_i, [_vp, _i, _i, _i, _i, _i]),
    "vda_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vda_forward_status": (_i, [_vp]),
    "vda_debug_copy": (_i, [_vp, C.c_char_p, _vp, C.c_int64, _vp]),
    "vda_set_option": (_i, [_vp, C.c_char_p, _i]),
    "vda_debug_occupy": (_i, [_i, _i, _ll, _vp]),
    "vda_profile_start": (_i, [_vp, _i]),
    "vda_profile_stop": (_i, [_vp, C.c_char_p, _i]),
}


class VdaError(RuntimeError):
    pass


def _load():
    # ONE HIP runtime per process: torch ships its own libamdhip64.so (soname libamdhip64.so.7, the one
    # libvda_hip.so asks for). Load torch's first so ours binds to it; loading ours first would pull in
    # /opt/rocm's copy as a second runtime whose streams and device state torch does not share.
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m video_depth_anything_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    runtimes = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    if len(runtimes) > 1:
        raise ImportError(f"two HIP runtimes in one process: {sorted(runtimes)}")
    tolerant = os.environ.get("VDA_LIB_TOLERANT") == "1"
    for name, (res, args) in SIGNATURES.items():
        if tolerant and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        raise VdaError(f"{what}: {lib.vda_last_error().decode()} (rc={rc})")
