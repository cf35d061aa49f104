"""ctypes binding of libskf.so (the C ABI declared in include/skf.h).

There is no CPU fallback: if the HIP library is missing or a call fails this
module raises.  PyTorch is used by callers only to own device memory/streams;
nothing here takes a torch type - pointers go through ``tensor.data_ptr()``.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libskf.so")

SKF_OK = 0
PREC_F32, PREC_BF16X3, PREC_BF16X6 = 0, 3, 6          # SKF_PREC_* of include/skf.h


def default_precision():
    """Arithmetic of the Dense matmuls when the caller does not choose: bf16x6 (exact 3-way split of the fp32 operands on
    the bf16 matrix cores - measured MORE accurate against float64 than the fp32-MFMA kernels, tests/test_gpu_ops.py);
    env SKF_GEMM_PRECISION = f32 | bf16x6 | bf16x3 overrides.  Read here, on the Python side: the library itself keeps
    no such state, every entry takes the mode as an argument / SkfConfig field."""
    e = os.environ.get("SKF_GEMM_PRECISION", "bf16x6").lower()
    try:
        return {"f32": PREC_F32, "bf16x6": PREC_BF16X6, "bf16x3": PREC_BF16X3, "0": 0, "3": 3, "6": 6}[e]
    except KeyError:
        raise ValueError("SKF_GEMM_PRECISION must be f32, bf16x6 or bf16x3 (got %r)" % e)


class SkfError(RuntimeError):
    pass


ATTN_TWO_PASS = 0x100                                 # SKF_ATTN_TWO_PASS
MODEL_DECODE_LAYERWISE = 1                            # SKF_MODEL_DECODE_LAYERWISE
MODEL_FFN_LAUNCHES = 2                                # SKF_MODEL_FFN_LAUNCHES: feed-forward blocks as separate launches
MODEL_TWO_STREAM_GRAPH = 4                            # SKF_MODEL_TWO_STREAM_GRAPH: opt-in for use_graph = 2 (HIP runtime fault, see skf.h)


class SkfConfig(C.Structure):
    """include/skf.h: struct SkfConfig, field for field (tests/test_cabi_cpu.py parses the header and compares).  struct_size is
    filled in by the constructor; the library refuses a config whose struct_size is not ITS sizeof(SkfConfig)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("batch", C.c_int32), ("seq_len", C.c_int32), ("d_model", C.c_int32), ("num_heads", C.c_int32),
        ("dff", C.c_int32), ("num_layers", C.c_int32),
        ("vocab_size", C.c_int32), ("n_classes", C.c_int32), ("lowerdim", C.c_int32), ("attn_version", C.c_int32),
        ("continuous", C.c_int32), ("blind_decoder_mask", C.c_int32), ("max_pos", C.c_int32),
        ("dropout_rate", C.c_float), ("recon_weight", C.c_float), ("class_weight", C.c_float),
        ("schedule", C.c_int32),
        ("sched_p0", C.c_float), ("sched_p1", C.c_float), ("sched_p2", C.c_float), ("sched_p3", C.c_float),
        ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("seed", C.c_uint32),
        ("use_graph", C.c_int32),
        ("optimizer", C.c_int32), ("momentum", C.c_float),
        ("class_buffer_layers", C.c_int32), ("class_dropout", C.c_float),
        ("do_classification", C.c_int32), ("do_reconstruction", C.c_int32),
        ("gemm_precision", C.c_int32), ("act_dtype", C.c_int32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if not self.struct_size:
            self.struct_size = C.sizeof(SkfConfig)


class SkfFfnBlockFwd(C.Structure):
    """include/skf.h: struct SkfFfnBlockFwd, field for field (skf_ffn_block_fwd_f32)."""
    _fields_ = [("struct_size", C.c_uint32), ("M", C.c_int32), ("d", C.c_int32), ("dff", C.c_int32), ("precision", C.c_int32),
                ("x", C.c_void_p), ("image", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p), ("h", C.c_void_p), ("relu_bits_out", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("z", C.c_void_p), ("out", C.c_void_p), ("stats", C.c_void_p),
                ("rate", C.c_float), ("site", C.c_uint32), ("step_state", C.c_void_p),
                ("pre_image", C.c_void_p), ("pre_bias", C.c_void_p), ("pre_residual", C.c_void_p), ("pre_gamma", C.c_void_p), ("pre_beta", C.c_void_p),
                ("pre_z", C.c_void_p), ("pre_out", C.c_void_p), ("pre_stats", C.c_void_p), ("pre_site", C.c_uint32), ("proj_n", C.c_int32),
                ("proj_image", C.c_void_p), ("proj_bias", C.c_void_p), ("proj_out", C.c_void_p)]


class SkfParamEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("row_stride", C.c_int32)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t
_U = C.c_uint

# name -> (restype, argtypes).  Mirrors include/skf.h one to one (tests check this).
SIGNATURES = {
    "skf_last_error": (C.c_char_p, []),
    "skf_version": (_I, []),
    "skf_device_info": (_I, [C.c_char_p, _Z, C.POINTER(_I)]),
    "skf_profiler_enable": (_I, [_I]),
    "skf_profiler_report": (_I, [C.c_char_p, _Z]),
    "skf_gemm_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "skf_gemm_default_splits": (_I, [_I, _I, _I]),
    "skf_gemm_wgrad_partial": (_I, [_I, _I, _I, _P, _I, _P, _I, _I, _I, _P, _Z, _P, _I, _P]),
    "skf_gemm_wgrad_partial_group": (_I, [_P, _I, _I, _P]),
    "skf_gemm_wgrad_partial_rows": (_I, [_I, _I, _I, _P, _I, _P, _I, _I, _I, _P, _Z, _P, _I, _P, _I, _P]),
    "skf_gemm_f32_rows": (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _Z, _I, _P, _I, _P]),
    "skf_gemm_relu_bits_bytes": (_Z, [_I, _I, _I, _I]),
    "skf_gemm_f32_bits": (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _Z, _I, _P, _I, _P, _P, _P]),
    "skf_gemm_ln_residual_supported": (_I, [_I, _I, _I, _I]),
    "skf_gemm_ln_residual_f32": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _U, _P, _I, _P]),
    "skf_ffn_fused_supported": (_I, [_I, _I, _I, _I]),
    "skf_ffn_image_bytes": (_Z, [_I, _I, _I]),
    "skf_ffn_relu_bits_bytes": (_Z, [_I, _I, _I, _I]),
    "skf_ffn_weight_images": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "skf_ffn_fused_fwd_f32": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U, _P, _I, _P]),
    "skf_ffn_fused_bwd_f32": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "skf_ffn_fused_ln_partials": (_I, [_I]),
    "skf_ffn_fused_bwd_ln_f32": (_I, [_I, _I, _I, _P, _P, _P, _P, _F, _U, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _I, _P]),
    "skf_dense_image_bytes": (_Z, [_I, _I, _I]),
    "skf_dense_weight_images": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "skf_layernorm_bwd_dgrad_supported": (_I, [_I, _I, _I]),
    "skf_layernorm_bwd_dgrad_partials": (_I, [_I]),
    "skf_layernorm_bwd_dgrad_f32": (_I, [_I, _I, _P, _P, _P, _P, _F, _U, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _I, _P]),
    "skf_layernorm_bwd_dgrad_lead_f32": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _F, _U, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _I, _P]),
    "skf_ffn_fused_fwd_proj_f32": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U, _P, _P, _P, _I, _P, _I, _P]),
    "skf_ffn_block_fwd_f32": (_I, [C.POINTER(SkfFfnBlockFwd), _P]),
    "skf_target_live_len": (_I, [_P, _I, _I, _I, _P, _P]),
    "skf_row_blocks_bytes": (_Z, [_I, _I]),
    "skf_row_blocks_build": (_I, [_P, _I, _I, _I, _P, _P]),
    "skf_splitk_reduce": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _I, _P]),
    "skf_splitk_reduce_blocks": (_I, [_I, _I]),
    "skf_splitk_reduce_batch": (_I, [_P, _I, _I, _P]),
    "skf_gemm_f32": (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _Z, _I, _P]),
    "skf_attention_fwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P]),
    "skf_attention_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I,
                               _P, _I, _P, _I, _P, _I, _I, _P]),
    "skf_attention_bwd_rows": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P]),
    "skf_attention_fwd_ordered": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P]),
    "skf_attention_bwd_ordered": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P]),
    "skf_sample_order": (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _P]),
    "skf_attention_weights": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "skf_attention_fwd_float_mask": (_I, [_P, _I, _P, _I, _P, _I, _P, C.c_long, C.c_long, C.c_long, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "skf_row_mean": (_I, [_P, _P, C.c_long, _I, _I, _P, _P]),
    "skf_embed_fwd": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _F, _U, _P, _P]),
    "skf_embed_bwd": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _F, _U, _P, _P]),
    "skf_embed_sort_workspace_bytes": (_Z, [_I, _I, _I]),
    "skf_embed_sort": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "skf_embed_bwd_sorted": (_I, [_P, _I, _I, _P, _I, _I, _P, _F, _U, _P, _P]),
    "skf_padding_mask": (_I, [_P, _I, _I, _I, _P, _P]),
    "skf_layernorm_residual_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _U, _P, _P]),
    "skf_layernorm_bwd_workspace_bytes": (_Z, [_I, _I]),
    "skf_layernorm_residual_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U, _P, _P, _Z, _P]),
    "skf_layernorm_residual_bwd_rows": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U, _P, _P, _Z, _P, _I, _P]),
    "skf_colsum": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "skf_softmax_ce": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P, _I, _P]),
    "skf_metrics_update": (_I, [_P, _P, _I, _F, _P, _P, _I, _F, _P, _P, _P]),
    "skf_padding_mask_continuous": (_I, [_P, _I, _I, _I, _P, _P]),
    "skf_embed_continuous_fwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _F, _U, _P, _P]),
    "skf_embed_continuous_bwd_workspace_bytes": (_Z, [_I, _I]),
    "skf_embed_continuous_bwd": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _F, _U, _P, _P, _Z, _P]),
    "skf_continuous_loss": (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _I, _P]),
    "skf_pool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "skf_pool_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "skf_expander_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "skf_expander_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _Z, _P]),
    "skf_config_size": (_Z, []),
    "skf_model_set_flags": (_I, [_P, C.c_uint32]),
    "skf_step_state_bytes": (_Z, []),
    "skf_step_prologue": (_I, [_P, _I, _F, _F, _F, _F, _F, _F, _U, _P]),
    "skf_step_epilogue": (_I, [_P, _P]),
    "skf_adam_step": (_I, [_P, _P, _P, _P, _Z, _P, _F, _F, _F, _F, _P]),
    "skf_sgd_momentum_step": (_I, [_P, _P, _P, _Z, _P, _F, _F, _P]),
    "skf_dropout": (_I, [_P, _P, _Z, _F, _U, _P, _P]),
    "skf_dropout_keep_mask": (_I, [_U, _U, _F, _Z, _P]),
    "skf_attention_decode": (_I, [_P, _I, _P, _P, _I, C.c_longlong, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I,
                                  _I, _P]),
    "skf_decode_init": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, C.c_longlong, _P, _P]),
    "skf_decode_embed": (_I, [_P, _P, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P]),
    "skf_decode_select_tokens": (_I, [_P, _I, _I, _I, _I, _I, C.c_longlong, _P, _I, _P, _I, _P, _P, _P, _P, _P]),
    "skf_decode_select_continuous": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P]),
    "skf_gemm_bf16": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P]),
    "skf_gemm_bf16_tile_rows": (_I, [_I, _I, _I, _I]),
    "skf_gemm_bf16_rows": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P]),
    "skf_gemm_bf16_relu_bits_bytes": (_Z, [_I, _I]),
    "skf_gemm_bf16_bits": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P]),
    "skf_gemm_bf16_wgrad_rows": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _Z, _P, _P]),
    "skf_gemm_bf16_wgrad_partial_rows": (_I, [_I, _I, _I, _P, _I, _P, _I, _I, _I, _P, _Z, _P, _P, _P]),
    "skf_gemm_bf16_wgrad_splits": (_I, [_I, _I, _I]),
    "skf_gemm_bf16_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "skf_gemm_bf16_wgrad_partial": (_I, [_I, _I, _I, _P, _I, _P, _I, _I, _I, _P, _Z, _P, _P]),
    "skf_gemm_bf16_wgrad": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _Z, _P]),
    "skf_attention_bf16_fwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "skf_attention_bf16_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "skf_attention_bf16_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I,
                                    _P, _I, _P, _I, _P, _I, _P, _Z, _P]),
    "skf_attention_bf16_bwd_rows": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _Z, _P, _P]),
    "skf_attention_bf16_fwd_ordered": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "skf_attention_bf16_bwd_ordered": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _Z, _P, _P, _P]),
    "skf_embed_fwd_bf16": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _F, _U, _P, _P]),
    "skf_embed_bwd_sorted_bf16": (_I, [_P, _I, _I, _P, _I, _I, _P, _F, _U, _P, _P]),
    "skf_layernorm_residual_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _U, _P, _P]),
    "skf_layernorm_bwd_bf16_workspace_bytes": (_Z, [_I, _I]),
    "skf_layernorm_residual_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U, _P, _P, _Z, _P]),
    "skf_layernorm_residual_bwd_bf16_rows": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U, _P, _P, _Z, _P, _I, _P]),
    "skf_softmax_ce_bf16": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P]),
    "skf_pool_fwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "skf_pool_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "skf_expander_fwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "skf_expander_bwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _Z, _P]),
    "skf_cast_weight_bf16": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _P]),
    "skf_cast_weight_bf16_blocks": (_I, [_I, _I, _I, _I, _P]),
    "skf_cast_weight_bf16_batch": (_I, [_P, _I, _I, _P]),
    "skf_cast_f32_to_bf16": (_I, [_P, _P, _Z, _P]),
    "skf_cast_bf16_to_f32": (_I, [_P, _P, _Z, _P]),
    "skf_config_validate": (_I, [C.POINTER(SkfConfig)]),
    "skf_model_param_floats": (_Z, [C.POINTER(SkfConfig)]),
    "skf_model_param_entries": (_I, [C.POINTER(SkfConfig), C.POINTER(SkfParamEntry), _I]),
    "skf_model_workspace_bytes": (_Z, [C.POINTER(SkfConfig)]),
    "skf_model_create": (_I, [C.POINTER(SkfConfig), C.POINTER(_P)]),
    "skf_model_destroy": (None, [_P]),
    "skf_model_bind": (_I, [_P, _P, _P, _P, _P, _P, _P, _Z, _P, _P]),
    "skf_model_forward": (_I, [_P, _P, _P, _I, _I, _P]),
    "skf_model_wait_inputs_staged": (_I, [_P, _P]),
    "skf_model_forward_backward": (_I, [_P, _P, _P, _I, _P, _P]),
    "skf_model_grad_buckets": (_I, [_P, _I, C.POINTER(_Z), C.POINTER(_Z)]),
    "skf_model_wait_grad_bucket": (_I, [_P, _I, _P]),
    "skf_model_apply_gradients_range": (_I, [_P, _Z, _Z, _F, _I, _P]),
    "skf_model_encode": (_I, [_P, _P, _P]),
    "skf_model_greedy_decode": (_I, [_P, _P, C.POINTER(_I), _I, C.c_longlong, C.c_longlong, _I, _P, C.POINTER(_I), _P]),
    "skf_model_apply_gradients": (_I, [_P, _F, _P]),
    "skf_model_buffer": (_I, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I)]),
    "skf_model_buffer_info": (_I, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
}

_lib = None


def load():
    """Load libskf.so; raises SkfError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SkfError("libskf.so not found at %s - run `python -m sketchformer_amd.build` "
                       "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64: import it first so that libskf.so binds to the SAME HIP runtime
    # (streams / device pointers are only meaningful inside one runtime instance).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != SKF_OK:
        lib = load()
        msg = lib.skf_last_error()
        raise SkfError("%s failed (rc=%d): %s" % (what or "libskf call", rc, msg.decode() if msg else ""))
    return rc


def call(name, *args):
    """Call an int-returning entry point and raise SkfError on a non-zero code."""
    lib = load()
    return check(getattr(lib, name)(*args), name)
