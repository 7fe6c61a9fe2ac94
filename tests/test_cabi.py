"""The C-ABI library loads and exports every symbol include/coponerf_hip.h declares (no compute: no GPU here)."""
import os

import pytest

from coponerf_amd import _hip


def test_library_is_built_and_exports_header_symbols():
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _hip.lib()
    declared = _hip.declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/coponerf_hip.h but not exported"
    assert set(_hip.SIGNATURES) | {"cpn_abi_version", "cpn_last_error", "cpn_gather_bwd_chunks", "cpn_conv_wgrad_scratch", "cpn_wgrad_tall_scratch", "cpn_conv4d_scratch", "cpn_conv4d_strided_bwd_scratch", "cpn_gn_stats_doubles", "cpn_scatter_tables_scratch", "cpn_encode_table_nodes", "cpn_encode_units", "cpn_linear_attention_scratch", "cpn_cost_volume_attention_scratch", "cpn_linear_attention_bwd_scratch", "cpn_cross_attention_bwd_scratch", "cpn_dwconv3x3_tokens_wgrad_scratch", "cpn_wgrad_f32_scratch_floats", "cpn_adam_chunk", "cpn_trunk_conv_scratch_floats", "cpn_device_cu_count", "cpn_stream_cu_count"} == set(declared)
    assert lib.cpn_abi_version() == _hip.ABI_VERSION


def test_argument_errors_are_status_codes_not_crashes():
    lib = _hip.lib()
    # null pointers are rejected before any launch (works without a GPU)
    assert lib.cpn_project_rays(None, None, 8, 1, 2, 4, None, None, None, None) == -1
    assert b"null" in lib.cpn_last_error()
    with pytest.raises(RuntimeError, match="cpn_gemm_f16"):
        _hip.call("cpn_gemm_f16", 16, 8, 16, 8, 16, 16, 8, 4, 100, 30, 0, 0, None)   # K not a multiple of 32
    # CU-masked streams: shares are multiples of 32 (8 XCDs x 4 shader engines); rejected before any HIP call
    import ctypes
    out = ctypes.c_void_p()
    assert lib.cpn_stream_create_cu_range(0, 200, ctypes.byref(out)) == -1 and b"multiples of 32" in lib.cpn_last_error()
    assert lib.cpn_stream_create_cu_range(0, 64, None) == -1
    assert lib.cpn_stream_destroy(None) == -1
    assert lib.cpn_stream_destroy(ctypes.c_void_p(1)) == -1 and b"not a stream" in lib.cpn_last_error()
    # round-3 get_z / training entry points: shape and pointer checks come before any launch
    assert lib.cpn_bn_act(None, None, None, None, None, None, 1e-5, 1, 8, 16, 1, None, None) == -1
    assert lib.cpn_bn_act(16, None, 16, 16, 16, 16, 1e-5, 1, 8, 6, 1, 16, None) == -2 and b"HW % 4" in lib.cpn_last_error()
    assert lib.cpn_bn_act(16, None, 16, 16, 16, 16, 1e-5, 1, 8, 16, 1, 20, None) == -1 and b"aligned" in lib.cpn_last_error()
    assert lib.cpn_corr_mean3(None, 16, None, 32, None, 64, 1, None, None) == -1
    assert lib.cpn_corr_mean3(16, 1, 16, 32, 16, 64, 1, 16, None) == -2
    assert lib.cpn_resize_bilinear_ac_adjoint(None, None, 1, 4, 4, 8, 8, None) == -1
    assert lib.cpn_resize_bilinear_ac_adjoint(16, 16, 0, 4, 4, 8, 8, None) == -2
    strided = lambda cin, cout, k, s: lib.cpn_conv4d_strided_bwd(16, 16, 16, 16, 1, cin, cout, 8, 8, 8, 8, k, s, (k - 1) // 2, 16,
                                                                 16, 16, 16, 16, None)
    assert strided(1, 4, 3, 2) == -2 and b"Cout = 8" in lib.cpn_last_error()
    assert strided(1, 8, 3, 1) == -2                                              # stride 1 has its own kernels
    assert strided(3, 8, 3, 2) == -2
    assert lib.cpn_conv4d_strided_bwd(16, 16, 16, 16, 1, 1, 8, 8, 8, 8, 8, 3, 2, 1, 16, 16, 16, None, None, None) == -1   # partial weight grads
    assert lib.cpn_conv4d_strided_bwd_scratch(1, 1, 8, 8, 8, 8, 8, 3, 1, 1) == 0
    assert lib.cpn_conv4d_strided_bwd_scratch(4, 1, 8, 64, 64, 64, 64, 5, 4, 2) > 4 * 2 * 64 * 64 * 16 * 16


def test_missing_library_raises(monkeypatch):
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libcoponerf_hip.so")
    with pytest.raises(_hip.HipLibraryError):
        _hip.lib()


def test_model_has_reference_parameter_names():
    """state_dict keys/shapes of the render path = SURVEY.md Appendix C.1 (checkpoint contract)."""
    from coponerf_amd import CoPoNeRF, synthetic as syn
    m = CoPoNeRF.CoPoNeRF(n_view=2)
    sd = m.state_dict()
    for name, shape in syn.RENDER_PARAM_SHAPES.items():
        assert name in sd and tuple(sd[name].shape) == shape, name
    assert tuple(sd["conv_map.weight"].shape) == (64, 3, 7, 7)
    assert m.npoints == 64 and CoPoNeRF.CoPoNeRF(n_view=2, npoints=0).npoints == 64
