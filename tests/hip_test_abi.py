"""ctypes bindings of the TEST-ONLY entry points of libegtr_hip.so (include/egtr_hip_test.h): explicit kernel-variant
selection for A/B parity tests (tests/test_gpu_kernels.py) and benchmarks (tools/msda_bench.py).  Deliberately NOT part of
the product binding (egtr_amd/_lib.py, egtr_amd/load_custom.py): the model never selects a variant."""
import ctypes

import torch

from egtr_amd import _lib
from egtr_amd.load_custom import _chk, _stream

_P, _I = ctypes.c_void_p, ctypes.c_int
SIGNATURES = {
    "egtr_msda_forward_f32_variant": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I],
    "egtr_msda_backward_f32_variant": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I],
    "egtr_test_decoder_drop_arrival": [_I],
    "egtr_test_l1_gather_buffer_bytes": [_I],
    "egtr_test_l1_gather_bandwidth": [_P, _P, _P, _I, _I, _P],
}


def _handle():
    h = _lib.lib()
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)
        fn.argtypes, fn.restype = argtypes, (ctypes.c_longlong if name.endswith("_bytes") else ctypes.c_int)
    return h


def msda_forward_variant(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, variant):
    """fp32 forward with an explicit kernel choice: 0 = automatic, 1 = wave-per-query, 3 = generic one-thread-per-element."""
    h = _handle()
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")):
        _chk(t, n, torch.float32)
    out = torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
    st = h.egtr_msda_forward_f32_variant(_stream(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                         sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, M, D, L, Lq, P,
                                         out.data_ptr(), variant)
    _lib.check(st, f"ms_deform_attn_forward(variant={variant})")
    return out


def msda_backward_variant(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, variant):
    """fp32 backward with an explicit kernel choice (include/egtr_hip_test.h)."""
    h = _handle()
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight"),
                 (grad_output, "grad_output")):
        _chk(t, n, torch.float32)
    grad_value = torch.zeros_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    st = h.egtr_msda_backward_f32_variant(_stream(), grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(),
                                          level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
                                          B, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(),
                                          grad_attn.data_ptr(), variant)
    _lib.check(st, f"ms_deform_attn_backward(variant={variant})")
    return grad_value, grad_loc, grad_attn


def decoder_drop_arrival(on):
    """Fault injection for egtr_decoder_layer_f32: while on, one wave of cluster 0 skips its second barrier arrival."""
    _lib.check(_handle().egtr_test_decoder_drop_arrival(1 if on else 0), "egtr_test_decoder_drop_arrival")
