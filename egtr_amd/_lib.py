"""ctypes binding of libegtr_hip.so (C ABI: include/egtr_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved, importing a kernel
entry point raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C egtr_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EGTR_HIP_LIBRARY: another build of the same library (same-box A/B of kernel builds: tools/*_bench.py, bench.py); there is
# still no fallback -- a path that does not exist raises like a missing in-tree build
LIB_PATH = os.environ.get("EGTR_HIP_LIBRARY") or os.path.join(_HERE, "libegtr_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
ABI_VERSION = 5   # include/egtr_hip.h: EGTR_ABI_VERSION of the header these signatures were written against

# name -> argtypes (restype is always int status unless listed in _RESTYPES)
SIGNATURES = {
    "egtr_abi_version": [],
    "egtr_status_string": [_I],
    "egtr_last_hip_error": [],
    "egtr_msda_forward_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "egtr_msda_forward_fused_vbias_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P,
                                          _P],
    "egtr_msda_forward_fused_box_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P,
                                        _P],
    "egtr_msda_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "egtr_msda_backward_out_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "egtr_msda_forward_f64": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "egtr_msda_backward_f64": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "egtr_msda_backward_bf16": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "egtr_msda_backward_bf16_workspace_floats": [_I, _I, _I, _I, _I, _I, _I],
    "egtr_msda_forward_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "egtr_msda_forward_fused_bf16": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    "egtr_self_attn_forward_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "egtr_self_attn_forward_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "egtr_self_attn_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "egtr_self_attn_backward_acc_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "egtr_linear_f32": [_P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, _I],
    "egtr_linear_grouped_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I],
    "egtr_linear_grouped_ln_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "egtr_add_layernorm_pos_f32": [_P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_float, _P, _I, _P],
    "egtr_bias_mask_rows_f32": [_P, _P, _P, _P, _I, _I, _I],
    "egtr_bias_relu_maxpool3x3s2_f32": [_P, _P, _P, _P, _I, _I, _I, _I],
    "egtr_box_decode_argmax_f32": [_P, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, _P, _P, _I, _P],
    "egtr_scale_rows_multi_f32": [_P, _I, _P, _P, _P, _P, _P],
    "egtr_bias_act_nchw_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I],
    "egtr_bias_act_nchw_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I],
    "egtr_bias_act_nhwc_bf16": [_P, _P, _P, _P, _P, ctypes.c_longlong, _I, _I],
    "egtr_bias_act_nhwc_f32": [_P, _P, _P, _P, _P, ctypes.c_longlong, _I, _I],
    "egtr_add_layernorm_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_float],
    "egtr_add_layernorm_pos_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_float, _P, _I, _P],
    "egtr_add_layernorm_f32": [_P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_float],
    "egtr_sine_pos_embed_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float],
    "egtr_level_geometry_f32": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P, _P, _P,
                                _P, _P],
    "egtr_level_geometry_bf16": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P, _P, _P,
                                 _P, _P],
    "egtr_input_proj_groupnorm_flatten_f32": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, _P, _P],
    "egtr_input_proj_groupnorm_flatten_bf16": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, _P, _P],
    "egtr_input_proj_groupnorm_tokens_bf16": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, _P, _P],
    "egtr_input_proj_groupnorm_tokens_f32": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, _P, _P],
    "egtr_input_proj_groupnorm_tokens_workspace_floats": [_I, _P, _I],
    "egtr_bbox_overlaps_f64": [_P, _P, _P, _I, _I, _I, _P],
    "egtr_hungarian_match_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, _I, ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P],
    "egtr_hungarian_match_scratch_doubles": [_I, _I, ctypes.c_longlong],
    "egtr_add_layernorm_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_float],
    "egtr_add_layernorm_backward_workspace_floats": [_I],
    "egtr_msda_geometry_forward_f32": [_P, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _I, _P, _P, _P, ctypes.c_longlong, _I, _I, _I],
    "egtr_msda_geometry_backward_f32": [_P, _P, _P, _P, _P, ctypes.c_longlong, _P, _I, _P, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _I, _I, _I],
    "egtr_linear_backward_f32": [_P, _P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _I, _I, _I],
    "egtr_linear_backward_acc_f32": [_P, _P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _I, _I, _I, _P, _P],
    "egtr_weighted_column_sum_f32": [_P, _P, _P, _P, _P, _I, _I],
    "egtr_column_sum_f32": [_P, _P, _P, _P, _P, _P, _I, _I],
    "egtr_column_sum_workspace_floats": [_I, _I],
    "egtr_any_nonfinite_f32": [_P, _P, ctypes.c_longlong, _P],
    "egtr_clamp_if_flag_f32": [_P, _P, _P, ctypes.c_longlong, _P, ctypes.c_float, _I],
    "egtr_detection_loss_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P, _P,
                                _P, _P],
    "egtr_relation_loss_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_float, _I, _I, _P, _P, _P, _P],
    "egtr_relation_loss_workspace_bytes": [_I, _I],
    "egtr_rel_head_forward_bf16w": [_P] * 16 + [_I] * 6 + [_P] * 3,
    "egtr_ffn_layernorm_bf16": [_P] * 7 + [ctypes.c_float, _P, _I, _P, _P, _I, _I, _I],
    "egtr_ffn_pack_weights_bf16": [_P, _P, _P, _I, _I, _P],
    "egtr_ffn_packed_weights_bytes": [_I],
    "egtr_linear_bf16": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_float],
    "egtr_rel_head_forward_bf16p": [_P] * 16 + [_I] * 6 + [_P] * 3,
    "egtr_rel_head_pack_tables_bf16": [_P, _P, _I, _I, _I, _P],
    "egtr_linear_split_bf16_f32": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I],
    "egtr_linear_split_bf16_wgrad_f32": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I],
    "egtr_linear_split_bf16_wgrad_workspace_floats": [_I, _I, _I],
    "egtr_gemm_split_tile_weights_f32": [_P, _P, _I, _I, _I, _I, _P],
    "egtr_gemm_split_tile_weights_pair_f32": [_P, _P, _I, _I, _I, _P],
    "egtr_gemm_split_tile_weights_multi_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "egtr_linear_split_bf16_grouped_pos_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "egtr_linear_split_bf16_ex_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "egtr_linear_split_bf16_wgrad_ex_f32": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _I, _P],
    "egtr_dropout_add_layernorm_f32": [_P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _I, _I, ctypes.c_float, _P],
    "egtr_dropout_add_layernorm_backward_workspace_floats": [_I],
    "egtr_dropout_add_layernorm_backward_f32": [_P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _P, ctypes.c_float, _P, _P, _P,
                                                _P, _I, _I, ctypes.c_float],
    "egtr_pad_batch_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "egtr_conv1x1_tail_x6_f32": [_P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I],
    "egtr_stem_conv7x7_pool_bf16": [_P, _P, _P, _P, _P, _I, _I, _I],
    "egtr_stem_conv7x7_pool_x6_f32": [_P, _P, _P, _P, _P, _I, _I, _I],
    "egtr_conv1x1_strided_x6_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "egtr_conv3x3_phase_channels": [_I, _I, _I, _I],
    "egtr_conv3x3_x6_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "egtr_conv1x1_tail_pack_weights_bf16": [_P, _P, _I, _I, _I, _P],
    "egtr_conv1x1_tail_bf16": [_P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I],
    "egtr_ffn_x6_f32": [_P, _P, _I, _P, _P, _P, _P, _P, _P, ctypes.c_float, _P, _I, _P, _P, _I, _I, _I],
    "egtr_encoder_tail_x6_f32": [_P, _P, _I, _P, _I, _P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P, ctypes.c_float,
                                 _P, _I, _P, _P, _I, _I, _I],
    "egtr_proj_ln_x6_f32": [_P, _P, _I, _P, _P, _P, _I, _P, _P, ctypes.c_float, _P, _I, _P, _P, _I, _I],
    "egtr_proj_multi_x6_f32": [_P, _P, _I, _P, _P, _P, _I, _I, _I],
    "egtr_xs_bytes": [_I, _I],
    "egtr_xs_split_f32": [_P, _P, _I, _P, _I, _I, _I, _P, _P, _I],
    "egtr_rel_head_forward_bf16x6_f32": [_P] * 16 + [_I] * 6 + [_P] * 3 + [_I],
    "egtr_rel_head_forward_bf16x6_save_f32": [_P] * 16 + [_I] * 6 + [_P] * 5,
    "egtr_rel_head_streams_f32": [_P] * 4 + [_I] * 2 + [_P] * 3,
    "egtr_rel_head_forward_save_f32": [_P] * 16 + [_I] * 6 + [_P] * 5,
    "egtr_rel_head_backward_pairs_f32": [_P] * 6 + [_I] * 4 + [_P] * 5,
    "egtr_decoder_layer_f32": [_P, _P],
    "egtr_decoder_layer_workspace": [_I, _I, _P, _P, _P],
}
_RESTYPES = {"egtr_status_string": ctypes.c_char_p, "egtr_last_hip_error": ctypes.c_char_p,
             "egtr_ffn_packed_weights_bytes": ctypes.c_longlong,
             "egtr_input_proj_groupnorm_tokens_workspace_floats": ctypes.c_longlong,
             "egtr_msda_backward_bf16_workspace_floats": ctypes.c_longlong,
             "egtr_hungarian_match_scratch_doubles": ctypes.c_longlong,
             "egtr_relation_loss_workspace_bytes": ctypes.c_longlong,
             "egtr_column_sum_workspace_floats": ctypes.c_longlong,
             "egtr_add_layernorm_backward_workspace_floats": ctypes.c_longlong,
             "egtr_linear_split_bf16_wgrad_workspace_floats": ctypes.c_longlong,
             "egtr_dropout_add_layernorm_backward_workspace_floats": ctypes.c_longlong,
             "egtr_xs_bytes": ctypes.c_longlong}

_lib = None


class EgtrHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises EgtrHipError if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EgtrHipError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `make -C egtr_amd/csrc` or __graft_entry__.build()). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        handle.egtr_abi_version.restype = ctypes.c_int
        got = handle.egtr_abi_version()
        if got != ABI_VERSION:   # a stale or foreign build: same symbol names, other contracts
            raise EgtrHipError(f"{LIB_PATH} has ABI version {got}, this package was written against {ABI_VERSION}: "
                               "rebuild it (make -C egtr_amd/csrc)")
        if os.environ.get("EGTR_HIP_LIBRARY"):
            import sys
            print(f"egtr_amd: EGTR_HIP_LIBRARY is set -- using {LIB_PATH} instead of the in-tree build", file=sys.stderr)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing -> loud
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        h = lib()
        msg = h.egtr_status_string(status).decode()
        if status == -2:
            msg += ": " + h.egtr_last_hip_error().decode()
        raise EgtrHipError(f"{what} failed: {msg} (status {status})")
