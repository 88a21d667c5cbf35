"""ctypes binding of libsfron.so (see include/sfron.h).  Fails loudly: no fallback path."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# tools/ may point LIB_PATH at another build of the same library (the debug-knob build, `make dbg`) BEFORE the first lib()
# call; the product reads no environment variable and has no fallback
LIB_PATH = os.path.join(_HERE, "libsfron.so")


class SfronError(RuntimeError):
    pass


_lib = None
ABI_VERSION = 16          # == sfron_abi_version() of the library these ctypes structs / prototypes were written for (checked on load)

_P = c_void_p     # device pointer
_S = c_void_p     # hipStream_t

_PROTOS = {
    "sfron_abi_version": (c_int, []),
    "sfron_build_arch": (c_char_p, []),
    "sfron_sweep_partials_len": (c_int, []),
    "sfron_sumsq_masked": (c_int, [_P, _P, _P, c_int64, _P, POINTER(c_int), _S]),
    "sfron_clip_coef": (c_int, [_P, c_int, c_float, _P, _S]),
    "sfron_masked_clip_adam": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double,
                                       c_double, _P, _P, c_double, c_int, _S]),
    "sfron_masked_clip_adam_wg": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double,
                                          c_double, _P, _P, c_double, c_int, c_int, _S]),
    "sfron_masked_clip_adam_q": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double, c_double, _P, _P,
                                         c_double, c_int, _P, _P, c_int, _S]),
    "sfron_sumsq_lowrank": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, POINTER(c_int), _S]),
    "sfron_adam_lowrank": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_double,
                                   c_double, _P, _P, c_double, c_int, _S]),
    "sfron_ema_update": (c_int, [_P, _P, c_int64, c_double, c_int, _S]),
    "sfron_fisher_accum": (c_int, [_P, _P, c_int64, c_float, _S]),
    "sfron_fisher_accum_clipped": (c_int, [_P, _P, _P, _P, c_int64, c_float, _S]),
    "sfron_mask_from_fisher": (c_int, [_P, _P, c_int64, c_float, _P, _S]),
    "sfron_cast_bf16": (c_int, [_P, _P, c_int64, _S]),
    "sfron_q_sample": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _S]),
    "sfron_dit_loss_fwd_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _S]),
    "sfron_p_sample": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _S]),
    "sfron_cfg_combine": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, _S]),
    "sfron_ddim_step": (c_int, [_P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, _P, _P, _S]),
    "sfron_ddpm_alphas_cumprod": (c_int, [_P, c_int, _P, _S]),
    "sfron_ddpm_q_sample": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _S]),
    "sfron_ddpm_sample_loss": (c_int, [_P, _P, c_int, c_int, _P, _S]),
    "sfron_ddpm_loss_coef": (c_int, [_P, c_int, c_int, c_float, c_float, c_int, _P, c_int, _P, _P, _S]),
    "sfron_ddpm_loss_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _S]),
}


class GemmDesc(ctypes.Structure):
    """Mirror of sfron_gemm_desc (include/sfron.h)."""
    _fields_ = [("A", c_void_p), ("B", c_void_p),
                ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int),
                ("a_transposed", c_int), ("b_transposed", c_int), ("epilogue", c_int), ("alpha", c_float),
                ("bias", c_void_p), ("c_bf16", c_void_p), ("ldc_bf16", c_int), ("c_f32", c_void_p), ("ldc_f32", c_int),
                ("aux", c_void_p), ("ldaux", c_int), ("gate", c_void_p), ("ldgate", c_int), ("pos", c_void_p),
                ("tokens", c_int), ("accumulate", c_int), ("resid", c_void_p), ("split_k", c_int),
                ("split_stride", ctypes.c_long), ("tile_hint", c_int), ("a_rowsum", c_void_p),
                ("rowsum_ws", c_void_p), ("col_partials", c_void_p), ("sumsq_mask", c_void_p), ("sumsq_partials", c_void_p)]


EPI_BF16, EPI_F32, EPI_GELU, EPI_GATE_RES, EPI_DGELU, EPI_POS = range(6)
EPI_GELU_Q, EPI_DGELU_Q = 7, 8          # round 6: GELU' as one byte per element (include/sfron.h)

_PROTOS["sfron_gemm_bf16"] = (c_int, [POINTER(GemmDesc), _S])
_PROTOS["sfron_gemm_rowsum_supported"] = (c_int, [c_int, c_int, c_int])
_PROTOS["sfron_gemm_gelu_q_supported"] = (c_int, [c_int, c_int, c_int])
_PROTOS["sfron_gemm_sumsq_partials"] = (c_int, [c_int, c_int, c_int])
_PROTOS["sfron_split_sum_bf16"] = (c_int, [_P, c_int, c_int64, c_int64, _P, _S])
_PROTOS["sfron_split_gate_res"] = (c_int, [_P, c_int, c_int64, _P, _P, c_int, c_int, _P, _P, _P, c_int, c_int, _S])
_PROTOS["sfron_sumsq_masked_ranges"] = (c_int, [_P, _P, _P, c_int, _P, _S])
_PROTOS["sfron_gemm_dgelu_colpart_rows"] = (c_int, [c_int, c_int, c_int])
_PROTOS.update({
    "sfron_rows_per_chunk": (c_int, [c_int]),
    "sfron_ln_modulate_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _S]),
    "sfron_ln_modulate_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _S]),
    "sfron_gate_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _S]),
    "sfron_ln_gate_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _S]),
    "sfron_reduce_chunks": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _S]),
    "sfron_weighted_reduce": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _S]),
    "sfron_reduce_slots": (c_int, [_P, ctypes.c_long, c_int, c_int, c_int, c_int, POINTER(c_void_p), POINTER(ctypes.c_long),
                                   POINTER(c_int), _S]),
    "sfron_split_finish": (c_int, [_P, c_int64, c_int, c_int, _P, _P, c_int, _S]),
    "sfron_groupnorm_one_launch": (c_int, [c_int, c_int, c_int, c_int]),
    "sfron_groupnorm_fwd_src": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_float, _P, _P, _P, _S]),
    "sfron_groupnorm_bwd_res_src": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, c_int, c_int, _P,
                                            c_int, _P, _P, _S]),
    "sfron_groupnorm_bwd_cast_src": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, _P, _P, _P, _S]),
    "sfron_reduce_batch": (c_int, [_P, c_int, _S]),
    "sfron_conv_wgrad_scatter_batch": (c_int, [_P, c_int, _S]),
    "sfron_reduce2": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _S]),
    "sfron_gated_bias_grads": (c_int, [_P, _P, c_int, ctypes.c_long, ctypes.c_long, c_int, c_int, c_int, _P, ctypes.c_long,
                                       ctypes.c_long, ctypes.c_long, _S]),
    "sfron_colsum": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _S]),
    "sfron_timestep_embed": (c_int, [_P, c_int, c_int, _P, c_int, _S]),
    "sfron_latent_sample": (c_int, [_P, _P, c_int, c_int, c_int, c_float, _P, _S]),
    "sfron_guard_inputs": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _S]),
    "sfron_guard_finite": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _S]),
    "sfron_silu_fwd": (c_int, [_P, c_int64, _P, _S]),
    "sfron_silu_bwd": (c_int, [_P, _P, c_int64, _P, _P, _S]),
    "sfron_cond_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _S]),
    "sfron_cond_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _S]),
    "sfron_patchify": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _S]),
    "sfron_unpatchify": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _S]),
    "sfron_attn_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _S]),
    "sfron_attn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _S]),
    "sfron_attn_bwd_form": (c_int, [c_int]),
    "sfron_attn_fwd_form": (c_int, [c_int]),
    "sfron_gemm_loader_waves": (c_int, [c_int]),
    "sfron_attn_bwd_bias_supported": (c_int, [c_int]),
    "sfron_attn_bwd_bias": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _S]),
})


class Fp8GemmDesc(ctypes.Structure):
    """Mirror of sfron_fp8_gemm_desc."""
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int), ("w_scale", c_void_p), ("a_scale", c_float),
                ("epilogue", c_int), ("bias", c_void_p), ("c_bf16", c_void_p), ("ldc_bf16", c_int), ("aux", c_void_p), ("ldaux", c_int),
                ("c_e4m3", c_void_p), ("c_e4m3_scale", c_float), ("c_f32", c_void_p), ("ldc_f32", c_int), ("resid", c_void_p),
                ("gate", c_void_p), ("ldgate", c_int), ("tokens", c_int), ("act_amax", c_void_p), ("aux_q", c_int)]


_PROTOS.update({
    "sfron_fp8_quant_tensors": (c_int, [_P, _P, c_int, _P, _P, _P, c_int, _S]),
    "sfron_fp8_update_scales": (c_int, [_P, c_int, _P, _S]),
    "sfron_cast_e4m3": (c_int, [_P, c_int, c_int64, c_float, _P, _P, _S]),
    "sfron_ln_modulate_fwd_q": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _S]),
    "sfron_fp8_gemm_supported": (c_int, [c_int, c_int, c_int]),
    "sfron_fp8_gemm": (c_int, [POINTER(Fp8GemmDesc), _S]),
})


class BGemmDesc(ctypes.Structure):
    """Mirror of sfron_bgemm_desc."""
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int),
                ("a_transposed", c_int), ("b_transposed", c_int), ("batch", c_int),
                ("stride_a", ctypes.c_long), ("stride_b", ctypes.c_long), ("stride_c", ctypes.c_long), ("batch2", c_int),
                ("stride_a2", ctypes.c_long), ("stride_b2", ctypes.c_long), ("stride_c2", ctypes.c_long), ("alpha", c_float),
                ("bias", c_void_p), ("c_bf16", c_void_p), ("c_f32", c_void_p), ("ldc", c_int), ("resid", c_void_p),
                ("sample_vec", c_void_p), ("ld_vec", c_int), ("rows_per_sample", c_int), ("accumulate", c_int),
                ("split_ws", c_void_p), ("split_ws_slabs", c_int)]


class ReduceItem(ctypes.Structure):
    """Mirror of sfron_reduce_item."""
    _fields_ = [("partials", c_void_p), ("out", c_void_p), ("groups", c_int), ("per_group", c_int), ("D", c_int), ("ldout", c_int)]


class WgradScatterItem(ctypes.Structure):
    """Mirror of sfron_wgrad_scatter_item."""
    _fields_ = [("dw_gemm", c_void_p), ("dw_oihw", c_void_p), ("slab_stride", c_int64), ("c_out", c_int), ("c_in", c_int), ("taps", c_int),
                ("c_in_p", c_int), ("n_slabs", c_int), ("reserved", c_int)]


class ConvDesc(ctypes.Structure):
    """Mirror of sfron_conv_desc."""
    _fields_ = [("batch", c_int), ("h_src", c_int), ("w_src", c_int), ("c_src", c_int), ("h_out", c_int), ("w_out", c_int),
                ("n_out", c_int), ("taps", c_int), ("stride", c_int), ("pad", c_int), ("upsample", c_int), ("dilate", c_int),
                ("bias", c_void_p), ("resid", c_void_p), ("sample_vec", c_void_p), ("ld_vec", c_int),
                ("out_bf16", c_void_p), ("out_f32", c_void_p), ("ld_out", c_int), ("accumulate", c_int),
                ("split_ws", c_void_p), ("split_ws_slabs", c_int), ("split_pending", POINTER(c_int))]


class SplitSrc(ctypes.Structure):
    """Mirror of sfron_split_src."""
    _fields_ = [("slabs", c_void_p), ("n_slabs", c_int), ("slab_stride", c_int64), ("bias", c_void_p), ("sample_vec", c_void_p), ("ld_vec", c_int),
                ("resid", c_void_p), ("ld_resid", c_int)]


_PROTOS.update({
    "sfron_bgemm_bf16": (c_int, [POINTER(BGemmDesc), _S]),
    "sfron_conv_fwd": (c_int, [POINTER(ConvDesc), _P, _P, _S]),
    "sfron_conv_wgrad": (c_int, [POINTER(ConvDesc), _P, c_int, _P, _P, _S]),
    "sfron_conv_wprep": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _S]),
    "sfron_conv_wgrad_splits": (c_int, [POINTER(ConvDesc)]),
    "sfron_conv_wgrad_scatter": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int64, _P, _S]),
    "sfron_nchw_to_rows_bf16": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _S]),
    "sfron_nchw_to_rows_f32": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _S]),
    "sfron_rows_to_nchw": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _S]),
    "sfron_groupnorm_fwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_float, _P, _P, _P, _P, _S]),
    "sfron_groupnorm_bwd": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, c_int, c_int,
                                    _P, _P, _P, _S]),
    "sfron_groupnorm_bwd_res": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, c_int, c_int,
                                        _P, c_int, _P, _P, _P, _S]),
    "sfron_groupnorm_bwd_cast": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, _P, _P, _P, _P, _S]),
    "sfron_groupnorm_bwd_cast_ok": (c_int, [c_int, c_int, c_int]),
    "sfron_groupnorm_chunks": (c_int, [c_int, c_int]),
    "sfron_groupnorm_scratch_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "sfron_softmax_fwd": (c_int, [_P, c_int64, c_int, c_int, c_float, _P, _S]),
    "sfron_layernorm_fwd": (c_int, [_P, _P, _P, c_int64, c_int, c_float, _P, _P, _P, _S]),
    "sfron_layernorm_rows_per_block": (c_int, [c_int64]),
    "sfron_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, c_int, _P, _P, _S]),
    "sfron_layernorm_bwd_res": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, c_int, _P, _P, _P, _S]),
    "sfron_geglu_fwd": (c_int, [_P, c_int64, c_int, _P, _S]),
    "sfron_geglu_bwd": (c_int, [_P, _P, c_int64, c_int, _P, _S]),
    "sfron_softmax_bwd": (c_int, [_P, _P, c_int64, c_int, c_float, _P, _S]),
    "sfron_sample_colsum": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int64, _S]),
    "sfron_axpby": (c_int, [_P, _P, c_float, c_float, c_int64, _P, _S]),
    "sfron_pool2_sum": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _S]),
    "sfron_cast_rows_bf16": (c_int, [_P, c_int, c_int64, c_int, _P, _S]),
    "sfron_cast_rows_colsum": (c_int, [_P, c_int, c_int64, c_int, _P, _P, c_int, _P, _S]),
    "sfron_cast_rows_colsum_partials": (c_int, [_P, c_int, c_int64, c_int, _P, _P, c_int, POINTER(c_int), _S]),
    "sfron_dropout_mask": (c_int, [ctypes.c_uint64, _P, c_int64, c_int64, c_float, _P, _S]),
    "sfron_dropout_mask_batch": (c_int, [ctypes.c_uint64, _P, _P, c_int, c_int64, c_float, _P, _S]),
    "sfron_conv_wprep_tiles": (c_int, [c_int, c_int]),
    "sfron_conv_wprep_batch": (c_int, [_P, c_int, c_int, _S]),
    "sfron_copy_cols": (c_int, [_P, c_int, c_int64, c_int, _P, c_int, c_int, _S]),
    "sfron_copy_cols2": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int64, _S]),
    "sfron_ddpm_timestep_embed": (c_int, [_P, c_int, c_int, _P, _S]),
    "sfron_class_embed_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _S]),
    "sfron_class_embed_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, _S]),
})


class DitCfg(ctypes.Structure):
    """Mirror of sfron_dit_cfg (include/sfron.h)."""
    _fields_ = [(n, c_int) for n in ("batch", "in_channels", "input_size", "patch", "hidden", "depth", "heads",
                                     "mlp_hidden", "num_classes", "freq_dim", "out_channels")]


DIT_LAYOUT_LEN = 24
_PROTOS.update({
    "sfron_dit_param_layout": (c_int, [POINTER(DitCfg), POINTER(c_int64), c_int]),
    "sfron_dit_workspace_bytes": (c_int64, [POINTER(DitCfg)]),
    "sfron_dit_sumsq_partials_len": (c_int, [POINTER(DitCfg)]),
    "sfron_dit_forward": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, _P, _P, _P, _P, _S]),
    "sfron_dit_backward": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, _P, _P, _P, _P, _S]),
    "sfron_dit_backward_dp": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _S]),
    "sfron_dit_scatter_late_bias": (c_int, [POINTER(DitCfg), _P, _P, _S]),
    "sfron_aux_create": (c_int, [POINTER(c_void_p)]),
    "sfron_aux_destroy": (c_int, [c_void_p]),
    "sfron_aux_wait_ada_factors": (c_int, [c_void_p, c_void_p]),
    "sfron_aux_streams": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p)]),
    "sfron_aux_set_probe": (c_int, [c_void_p, c_void_p]),
    "sfron_aux_arm_sumsq": (c_int, [c_void_p, _P, _P]),
    "sfron_aux_wait_ada": (c_int, [c_void_p, _S]),
    "sfron_fp8_activation_amax": (c_int, [_P, POINTER(c_float), c_int, c_void_p]),
    "sfron_dit_forward_probed": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _S]),
    "sfron_dit_forward_after": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _S]),
    "sfron_dit_forward_phase": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _S]),
    "sfron_dit_forward_fp8_phase": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, POINTER(c_float), _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _S]),
    "sfron_dit_fp8_workspace_bytes": (c_int64, [POINTER(DitCfg)]),
    "sfron_dit_forward_fp8": (c_int, [POINTER(DitCfg), _P, _P, _P, _P, POINTER(c_float), _P, _P, _P, _P, _P, _P, _P, _P, _P, _S]),
    "sfron_probe_create": (c_int, [c_int, POINTER(c_void_p)]),
    "sfron_probe_reset": (c_int, [c_void_p]),
    "sfron_probe_read": (c_int, [c_void_p, POINTER(c_int), POINTER(c_double)]),
    "sfron_probe_destroy": (c_int, [c_void_p]),
})


class WprepItem(ctypes.Structure):        # sfron_wprep_item
    _fields_ = [("w", c_void_p), ("fwd", c_void_p), ("dgr", c_void_p), ("co", ctypes.c_int32), ("ci", ctypes.c_int32), ("co_p", ctypes.c_int32),
                ("ci_p", ctypes.c_int32), ("tile0", ctypes.c_int32), ("pad_", ctypes.c_int32)]


def declared_symbols():
    return sorted(_PROTOS)


def lib():
    """Load (once) and return the ctypes handle; raise SfronError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise SfronError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             f"(make -C {os.path.join(_HERE, 'csrc')}). There is no fallback path.")
        # torch's HIP runtime first: the library must bind to the runtime instance torch uses, and torch must be the one that
        # initialises the device (the other order leaves the library's first stream / event creation without a device: status 100)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(h, name)      # AttributeError if the symbol is missing: loud by design
            fn.restype, fn.argtypes = res, args
        got = h.sfron_abi_version()
        if got != ABI_VERSION:      # a stale build, or a variant library built from other sources: structs would be read past their end
            raise SfronError(f"{LIB_PATH} reports ABI version {got}, this package's ctypes structs were written for {ABI_VERSION}: rebuild it "
                             f"(python -c 'import __graft_entry__ as g; g.build()')")
        _lib = h
    return _lib


def check(status, what):
    if status != 0:
        raise SfronError(f"{what} failed with status {status}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Tensors must be contiguous and on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SfronError("sfron ops need GPU tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise SfronError("sfron ops need contiguous tensors")
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
