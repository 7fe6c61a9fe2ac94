"""ctypes binding of libcoponerf_hip.so (C ABI: include/coponerf_hip.h).

The library is the product's only compute path: if it is missing or a symbol
is absent this module raises — there is no PyTorch/CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
# COPONERF_HIP_LIB: another build of the SAME library (kernel-variant experiments, tools/_build/); never a fallback
LIB_PATH = os.environ.get("COPONERF_HIP_LIB") or os.path.join(_HERE, "libcoponerf_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "coponerf_hip.h")

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argtypes (every function returns int status except the two noted below)
SIGNATURES: Dict[str, List] = {
    "cpn_stream_create_cu_range": [_I, _I, ctypes.POINTER(ctypes.c_void_p)],
    "cpn_stream_destroy": [_P],
    "cpn_project_rays": [_P, _P, ctypes.c_longlong, _I, _I, _I, _P, _P, _P, _P],
    "cpn_sample_geometry": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "cpn_nchw_to_nhwc_f16": [_P, _P, _I, _I, _I, _I, _P],
    "cpn_pack_weight_f16": [_P, _I, _I, _P, _I, _P],
    "cpn_pack_encode_weights": [_P, _I, _P, _P, _P],
    "cpn_node_features": [_P, _P, _P, _I, _I, _I, _P, _P],
    "cpn_encode_key": [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "cpn_local_units": [_I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cpn_local_hidden": [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "cpn_gemm_f16": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "cpn_pack_gemm_frags": [_P, _I, _I, _I, _P, _P],
    "cpn_gemm_f16_fewrows": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cpn_attend_hidden": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "cpn_node_features_f32": [_P, _P, _P, _I, _I, _I, _P, _P],
    "cpn_encode_hidden_f32": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "cpn_attend_hidden_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "cpn_linear_f32": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "cpn_lightfield_decode": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "cpn_ray_outputs": [_P, _P, _P, ctypes.c_longlong, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "cpn_attend_hidden_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "cpn_hid_grad_combine": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "cpn_gemm_f16_combine": [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "cpn_gemm_f16_masked": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "cpn_wgrad_skinny_f16": [_P, _P, _I, ctypes.c_longlong, _P, _P, _P],
    "cpn_wgrad_tall_f16": [_P, _I, _P, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P],
    "cpn_local_hidden_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "cpn_gather_rows_bwd_level3": [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "cpn_scatter_rows_tables": [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "cpn_node_features_bwd": [_P, _I, _I, _I, _P, _P, _P, _P],
    "cpn_scale_to_f16": [_P, ctypes.c_longlong, _F, _P, _P, _P, _P],
    "cpn_gather_tail": [_P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "cpn_conv4d_gn_relu": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cpn_conv4d": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cpn_transpose_pairs": [_P, _I, _I, _I, _P, _P],
    "cpn_conv4d_dgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "cpn_gn_relu": [_P, _P, _P, _P, _P, _F, _I, _I, ctypes.c_longlong, _P, _P],
    "cpn_gn_relu_bwd": [_P, _P, _P, _P, _P, _F, _I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P],
    "cpn_conv_wgrad_planes": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cpn_dwconv3x3_wgrad": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "cpn_dwconv3x3_tokens": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "cpn_dwconv3x3_tokens_wgrad": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "cpn_dual_softmax": [_P, _I, _I, _I, _P, _P, _P, _P],
    "cpn_dual_softmax_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "cpn_correlation": [_P, _P, _I, _I, _I, _F, _P, _P, _P, _P],
    "cpn_l2norm_rows_bwd": [_P, _P, _P, ctypes.c_longlong, _I, _F, _P, _P],
    "cpn_wgrad_f32": [_P, _I, _P, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P],
    "cpn_adam_step": [_P, _P, _I, _P, _P, _P, _P, _P, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P],
    "cpn_soft_argmax_pair": [_P, _I, _I, _F, _P, _P, _P],
    "cpn_soft_argmax_pair_bwd": [_P, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "cpn_linear_attention": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P],
    "cpn_cross_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "cpn_linear_attention_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    "cpn_qk_assemble": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "cpn_cost_volume_attention": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P],
    "cpn_cross_attention": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "cpn_conv_map7x7": [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    "cpn_pose_positional": [_P, _I, _I, _F, _P, _I, _P, _P],
    "cpn_pose_gemv": [_P, _P, _I, _I, _I, _I, _P, _P],
    "cpn_pose_tail": [_P, _I, _P, _P, _I, _P, _P],
    "cpn_pack_conv_weight": [_P, _I, _I, _I, _P, _P],
    "cpn_trunk_conv_bn_act": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _I, _P, _P, _P, _P],
    "cpn_bn_act": [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P, _P],
    "cpn_prepare_input": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cpn_resize_bilinear_ac": [_P, _P, ctypes.c_longlong, _I, _I, _I, _I, _P],
    "cpn_resize_bilinear_ac_adjoint": [_P, _P, ctypes.c_longlong, _I, _I, _I, _I, _P],
    "cpn_corr_mean3": [_P, _I, _P, _I, _P, _I, _I, _P, _P],
    "cpn_conv4d_strided_bwd": [_P, _P, _P, _P] + [_I] * 10 + [_P] * 6,
}

CAM_STRIDE = 96
CAM_TQ, CAM_M, CAM_AOWN, CAM_AOTH, CAM_KQ, CAM_KC, CAM_KO, CAM_KN = 0, 16, 32, 48, 64, 68, 72, 76
TAB_LD = 832
RAYC_STRIDE = 64
LIGHTFIELD_PACK_FLOATS = 128 * 32 + 128 + 3 * (128 * 416 + 128 + 2 * (128 * 128 + 128)) + 16 * 128 + 16
ABI_VERSION = 10
ADAM_SEG_BYTES = 48


def declared_symbols() -> List[str]:
    """Every function name include/coponerf_hip.h declares."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(cpn_[a-z0-9_]+)\s*\(", text)))


class HipLibraryError(RuntimeError):
    pass


_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) and type the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python coponerf_amd/csrc/build.py` "
            "(or __graft_entry__.build()).  coponerf_amd has no non-HIP compute path.")
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    handle.cpn_abi_version.argtypes = []
    handle.cpn_abi_version.restype = ctypes.c_int
    handle.cpn_device_cu_count.argtypes = []
    handle.cpn_device_cu_count.restype = ctypes.c_int
    handle.cpn_stream_cu_count.argtypes = [_P]
    handle.cpn_stream_cu_count.restype = ctypes.c_int
    handle.cpn_encode_table_nodes.argtypes = [_I, _I]
    handle.cpn_encode_table_nodes.restype = ctypes.c_longlong
    handle.cpn_encode_units.argtypes = [_I] * 5
    handle.cpn_encode_units.restype = ctypes.c_longlong
    handle.cpn_linear_attention_scratch.argtypes = [_I, _I, _I, _I]
    handle.cpn_linear_attention_scratch.restype = ctypes.c_longlong
    handle.cpn_cost_volume_attention_scratch.argtypes = [_I] * 5
    handle.cpn_cost_volume_attention_scratch.restype = ctypes.c_longlong
    handle.cpn_cross_attention_bwd_scratch.argtypes = [_I, _I, _I, _I]
    handle.cpn_cross_attention_bwd_scratch.restype = ctypes.c_longlong
    handle.cpn_linear_attention_bwd_scratch.argtypes = [_I, _I, _I, _I, _I]
    handle.cpn_linear_attention_bwd_scratch.restype = ctypes.c_longlong
    handle.cpn_gather_bwd_chunks.argtypes = [_I, _I]
    handle.cpn_gather_bwd_chunks.restype = ctypes.c_longlong
    handle.cpn_scatter_tables_scratch.argtypes = [_I] * 6
    handle.cpn_scatter_tables_scratch.restype = ctypes.c_longlong
    handle.cpn_conv4d_strided_bwd_scratch.argtypes = [_I] * 10
    handle.cpn_conv4d_strided_bwd_scratch.restype = ctypes.c_longlong
    handle.cpn_conv4d_scratch.argtypes = [_I] * 7
    handle.cpn_conv4d_scratch.restype = ctypes.c_longlong
    handle.cpn_gn_stats_doubles.argtypes = [_I, _I, ctypes.c_longlong]
    handle.cpn_gn_stats_doubles.restype = ctypes.c_longlong
    handle.cpn_wgrad_tall_scratch.argtypes = [_I, _I]
    handle.cpn_wgrad_tall_scratch.restype = ctypes.c_longlong
    handle.cpn_conv_wgrad_scratch.argtypes = [_I, _I]
    handle.cpn_conv_wgrad_scratch.restype = ctypes.c_longlong
    handle.cpn_dwconv3x3_tokens_wgrad_scratch.argtypes = [_I, _I, _I]
    handle.cpn_dwconv3x3_tokens_wgrad_scratch.restype = ctypes.c_longlong
    handle.cpn_wgrad_f32_scratch_floats.argtypes = [ctypes.c_longlong, _I, _I]
    handle.cpn_wgrad_f32_scratch_floats.restype = ctypes.c_longlong
    handle.cpn_trunk_conv_scratch_floats.argtypes = [_I] * 7
    handle.cpn_trunk_conv_scratch_floats.restype = ctypes.c_longlong
    handle.cpn_adam_chunk.argtypes = []
    handle.cpn_adam_chunk.restype = ctypes.c_int
    handle.cpn_last_error.argtypes = []
    handle.cpn_last_error.restype = ctypes.c_char_p
    got = handle.cpn_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"{LIB_PATH} has ABI version {got}, binding expects {ABI_VERSION}; rebuild it")
    _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().cpn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")


_FN = {}


def call(name: str, *args) -> None:
    """Invoke an entry point; tensors are passed as .data_ptr() integers by the caller."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    status = fn(*args)
    if status != 0:
        check(status, name)


def stream_handle() -> int:
    """Raw handle of torch's current HIP stream on the current device.  The same lookup as
    `torch.cuda.current_stream().cuda_stream` without building a Stream object: that form cost 9 us of Python per launch —
    1.3 ms of the 6.5 ms of host time of one get_z, which had become the eager call's bound."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
