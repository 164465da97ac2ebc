"""ctypes binding of libsdtrain_hip.so (include/sdt.h).  The product path has no CPU fallback: if the
library is missing, or no HIP device is visible when a kernel is requested, it raises."""
import ctypes
import os

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float
_D = ctypes.c_double


class SdtConvGeom(ctypes.Structure):
    _fields_ = [(n, _I) for n in ("batch", "in_h", "in_w", "out_h", "out_w", "kh", "kw", "stride", "pad_top", "pad_left")]


class SdtAttnDesc(ctypes.Structure):
    _fields_ = [("B", _I), ("H", _I), ("Nq", _I), ("Nk", _I), ("D", _I), ("ldq", _I), ("ldk", _I), ("ldv", _I),
                ("ldo", _I), ("scale", _F), ("causal", _I), ("ldgrad_q", _I), ("ldgrad_k", _I), ("ldgrad_v", _I),
                ("ld_dout", _I), ("key_weight", _P)]


class SdtPrepDesc(ctypes.Structure):
    _fields_ = [("src_off", _L), ("w_off", _L), ("wt_off", _L), ("batch", _I), ("R", _I), ("C", _I), ("Rp", _I),
                ("Cp", _I), ("tile0", _I), ("flags", _I)]


class SdtNormGradJob(ctypes.Structure):
    _fields_ = [("partial", _P), ("dgamma", _P), ("dbeta", _P), ("nrows", _I), ("C", _I)]


class SdtConvWgradProblem(ctypes.Structure):
    _fields_ = [("A", _P), ("dY", _P), ("dW", _P), ("dbias", _P), ("geom", SdtConvGeom), ("K1", _I), ("N", _I), ("K1_valid", _I),
                ("N_valid", _I), ("lda", _I), ("ldb", _I), ("sq_slots", _P), ("dw_bf16", _I)]


class SdtTnProblem(ctypes.Structure):
    _fields_ = [("A", _P), ("dY", _P), ("dW", _P), ("dbias", _P), ("M", _L), ("K1", _I), ("N", _I), ("K1_valid", _I),
                ("N_valid", _I), ("lda", _I), ("ldb", _I), ("ldw", _I), ("n_seg", _I), ("seg_stride", _L), ("sq_slots", _P), ("dw_bf16", _I)]


GATHER_PLAIN, GATHER_FPROP, GATHER_DGRAD = 0, 1, 2
ACT_SILU, ACT_QUICK_GELU, ACT_GELU_ERF = 0, 1, 2

# name -> argtypes (all return int); the trailing stream argument is included
SIGNATURES = {
    "sdt_add_noise_velocity": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sdt_vae_posterior_sample": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "sdt_ddim_cfg_step": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _I, _P],
    "sdt_mse_loss_fwd_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P],
    "sdt_timestep_embedding": [_P, _P, _I, _I, _I, _F, _P],
    "sdt_sqnorm_accumulate": [_P, _L, _P, _P, _L, _P],
    "sdt_sqnorm_accumulate_bf16": [_P, _L, _P, _P, _L, _P],
    "sdt_lion8_step": [_P, _P, _I, _P, _P, _P, _P, _L, _I, _P, _P, _D, _D, _D, _D, _D, _D, _P],
    "sdt_lion32_step": [_P, _P, _P, _P, _P, _L, _P, _D, _D, _D, _D, _D, _D, _P],
    "sdt_lion8_quantize": [_P, _P, _P, _L, _I, _P, _P],
    "sdt_lion8_dequantize": [_P, _P, _P, _L, _I, _P],
    "sdt_groupnorm_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _I, _P, _L, _P],
    "sdt_groupnorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _L, _P],
    "sdt_layernorm_fwd": [_P, _P, _P, _P, _P, _L, _I, _F, _P],
    "sdt_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _L, _P],
    "sdt_norm_param_grads_group": [_P, _I, _P],
    "sdt_sum_n_bf16": [_P, _I, _P, _L, _P],
    "sdt_event_create": [_P],
    "sdt_event_destroy": [_P],
    "sdt_event_record": [_P, _I, _P],
    "sdt_stream_wait_event": [_P, _P],
    "sdt_stream_wait_event_external": [_P, _P],
    "sdt_gemm_nt_bf16": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _L, _I, _I, _I, _I, _P, _P, _L, _P, _I, _I, _I, _L, _I, _P],
    "sdt_gemm_nt_gn_parts": [_L, _I, _I, _I, _I, _I, _I, _P],
    "sdt_gemm_tn_wgrad": [_P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _L, _I, _P, _P, _L, _P, _P],
    "sdt_sum_f64_accumulate": [_P, _L, _P, _P, _L, _P],
    "sdt_gemm_tn_wgrad_group": [_P, _I, _P, _L, _P],
    "sdt_conv_wgrad_group": [_P, _I, _P, _L, _P],
    "sdt_zero_ranges": [_P, _P, _I, _P],
    "sdt_colsum_accumulate": [_P, _P, _L, _I, _I, _P, _L, _P],
    "sdt_colsum_batched_bf16": [_P, _P, _I, _L, _I, _I, _P, _L, _P],
    "sdt_attention_fwd": [_P, _P, _P, _P, _P, _P, _P],
    "sdt_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "sdt_softmax_rows_inplace": [_P, _L, _I, _F, _P],
    "sdt_act_fwd": [_P, _P, _L, _I, _P],
    "sdt_act_bwd": [_P, _P, _P, _L, _I, _P],
    "sdt_geglu_fwd": [_P, _P, _L, _I, _P],
    "sdt_geglu_bwd": [_P, _P, _P, _L, _I, _P],
    "sdt_copy2d_bf16": [_P, _L, _P, _L, _L, _I, _P],
    "sdt_copy_cols_bf16": [_P, _L, _P, _P, _P, _I, _L, _I, _P],
    "sdt_add_bf16": [_P, _P, _P, _L, _P],
    "sdt_upsample2x_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "sdt_upsample2x_bwd": [_P, _P, _I, _I, _I, _I, _P],
    "sdt_nchw_f32_to_nhwc_bf16": [_P, _P, _I, _I, _I, _I, _I, _P],
    "sdt_nhwc_bf16_to_nchw_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "sdt_cast_f32_to_bf16": [_P, _P, _L, _P],
    "sdt_transpose_bf16": [_P, _P, _I, _I, _I, _P],
    "sdt_param_prepare": [_P, _P, _P, _P, _I, _I, _P],
    "sdt_embedding_fwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "sdt_embedding_bwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "sdt_ff_geglu_fwd": [_P, _P, _P, _P, _P, _L, _I, _I, _P],
    "sdt_ff_geglu_supported": [_L, _I, _I],
}
WS_QUERY = {"sdt_gemm_nt_workspace_bytes": [_L, _I, _I, _I], "sdt_gemm_tn_workspace_bytes": [_L, _I, _I, _I, _I, _I, _P], "sdt_layernorm_bwd_workspace_bytes": [_L, _I], "sdt_layernorm_bwd_partial_rows": [_L, _I],
            "sdt_groupnorm_bwd_workspace_bytes": [_I, _I, _I],
            "sdt_groupnorm_fwd_workspace_bytes": [_I, _I, _I, _I], "sdt_attention_bwd_workspace_bytes": [_P],
            "sdt_gemm_tn_wgrad_group_workspace_bytes": [_P, _I], "sdt_conv_wgrad_group_workspace_bytes": [_P, _I], "sdt_reduce_workspace_bytes": [], "sdt_sqnorm_workspace_bytes": [], "sdt_colsum_workspace_bytes": [_I, _L, _I],
            "sdt_wgrad_sq_slots": [_I, _I, _I]}
NOARG = {"sdt_abi_version": _I, "sdt_gemm_tn_wgrad_group_max": _I, "sdt_norm_param_grads_group_max": _I, "sdt_zero_ranges_chunk": _I, "sdt_device_count": _I, "sdt_param_prepare_desc_size": _I, "sdt_last_error": ctypes.c_char_p}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsdtrain_hip.so")
if os.environ.get("SDT_LIB"):  # developer A/B of two builds on one box (tools/ab_libs.sh): another in-tree build of the same sources
    LIB_PATH = os.path.abspath(os.environ["SDT_LIB"])
_lib = None


class SdtError(RuntimeError):
    pass


def load():
    """Load the HIP library (never a fallback).  Raises SdtError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdtError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: it ships its own HIP runtime (libamdhip64), and the library must bind to THAT copy - loaded the other
    # way round, /opt/rocm's runtime comes in beside torch's and torch then finds no device
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _I
    for name, argtypes in WS_QUERY.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _L
    for name, res in NOARG.items():
        fn = getattr(lib, name)
        fn.argtypes = []
        fn.restype = res
    if lib.sdt_param_prepare_desc_size() != ctypes.sizeof(SdtPrepDesc):
        raise SdtError("SdtPrepDesc layout mismatch between _lib.py and the library")
    _lib = lib
    return lib


def require_device():
    lib = load()
    if lib.sdt_device_count() < 1:
        raise SdtError("no HIP device visible: the train_step hot path runs only on the HIP kernels (no CPU fallback)")
    return lib


def call(name, *args):
    lib = _lib if _lib is not None else load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SdtError(f"{name} failed ({rc}): {lib.sdt_last_error().decode()}")
