"""The C-ABI library builds for gfx950, loads, and exports every symbol include/lsq_hip.h declares.
No compute calls here (no GPU in the build container); argument validation paths are exercised
because they return before any launch."""

import ctypes
import os
import re

import pytest
import torch  # noqa: F401  (its HIP runtime must be the one the library binds to)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'lsq_hip.h')


DEBUG_HEADER = os.path.join(ROOT, 'include', 'lsq_hip_debug.h')


def declared_functions(header=HEADER):
    text = re.sub(r'/\*.*?\*/', '', open(header).read(), flags=re.S)
    return sorted(set(re.findall(r'\b(lsq_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def hip():
    from quant import _hip
    if not _hip.available():
        import __graft_entry__
        __graft_entry__.build()
    return _hip


def test_header_declares_the_expected_entry_points():
    assert declared_functions() == sorted([
        'lsq_abi_version', 'lsq_error_string', 'lsq_act_plane_words', 'lsq_weight_plane_words', 'lsq_solver_workspace_bytes', 'lsq_sweep_workspace_bytes',
        'lsq_act_quant', 'lsq_solve_rows', 'lsq_pack_weight', 'lsq_xnor_conv2d', 'lsq_signw_conv2d', 'lsq_signw_weight_bytes', 'lsq_signw_prepare_weight',
        'lsq_pool_bias_relu_nhwc', 'lsq_stem_conv_pool', 'lsq_pointwise_conv', 'lsq_quant_values', 'lsq_ste_backward', 'lsq_xnor_conv2d_chain',
        # ABI v11: activation tensors in the three-stream row layout
        'lsq_split3_stream_floats', 'lsq_layout_support', 'lsq_act_quant_layout', 'lsq_xnor_conv2d_layout'])


def test_every_exported_symbol_is_declared_in_a_header(hip):
    """The product ABI (lsq_hip.h) and the test hooks (lsq_hip_debug.h) together are ALL the library exports under the
    lsq_ prefix: nothing undeclared can change what a call does."""
    import shutil
    import subprocess
    nm = shutil.which('nm')
    if nm is None:
        pytest.skip('no nm on this machine')
    out = subprocess.run([nm, '-D', '--defined-only', hip.library_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if ' T ' in line and line.split()[-1].startswith('lsq_')})
    assert declared_functions(DEBUG_HEADER) == ['lsq_debug_force_streaming', 'lsq_debug_fused_mode', 'lsq_debug_solver_trace', 'lsq_debug_xnor_impl']
    assert exported == sorted(declared_functions() + declared_functions(DEBUG_HEADER)), exported


def test_debug_switches_restore_the_previous_values(hip):
    lib = hip.lib()
    assert lib.lsq_debug_xnor_impl(0) == 0 and lib.lsq_debug_force_streaming(0) == 0 and lib.lsq_debug_fused_mode(0) == 0
    with pytest.raises(RuntimeError):
        with hip.debug_switches(xnor_popcount=True, force_streaming=True, fused_mode=2):
            assert lib.lsq_debug_xnor_impl(1) == 1 and lib.lsq_debug_fused_mode(2) == 2
            raise RuntimeError('a failing test body')
    assert lib.lsq_debug_xnor_impl(0) == 0 and lib.lsq_debug_force_streaming(0) == 0 and lib.lsq_debug_fused_mode(0) == 0


def test_library_exports_every_declared_symbol(hip):
    lib = hip.lib()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.lsq_abi_version() == 11
    assert lib.lsq_error_string(0) == b'ok'
    assert b'NULL' in lib.lsq_error_string(-1)


def test_geometry_helpers_and_argument_errors(hip):
    lib = hip.lib()
    g = hip.make_geom(256, 64, 56, 56, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    assert lib.lsq_act_plane_words(ctypes.byref(g)) == 256 * 1 * 58 * 58
    assert lib.lsq_weight_plane_words(ctypes.byref(g)) == 9 * 1 * 64
    g2 = hip.make_geom(2, 20, 12, 12, 50, 5, 5, (1, 1), (0, 0), (1, 1), 1)
    assert lib.lsq_act_plane_words(ctypes.byref(g2)) == 2 * 1 * 12 * 12
    assert lib.lsq_weight_plane_words(ctypes.byref(g2)) == 25 * 1 * 64          # 50 -> padded to 64
    assert hip.out_hw(g2) == (8, 8)
    bad = hip.make_geom(2, 20, 12, 12, 50, 5, 5, (1, 1), (0, 0), (1, 1), 3)      # C % groups != 0
    assert lib.lsq_act_plane_words(ctypes.byref(bad)) == -1
    # null pointers / bad schemes are rejected before any launch
    assert lib.lsq_act_quant(None, ctypes.byref(g), 1, 1, 3, 2.0, None, None, None, None, None, None, 0, None) == -1
    assert lib.lsq_solve_rows(None, 1, 1, 1, 0, -1.0, None, None, None, 0, None) == -1
    assert lib.lsq_xnor_conv2d(None, 1, None, None, None, 1, None, None, ctypes.byref(g), 0, None, None, None, None, None) == -1


def test_cuda_path_has_no_fallback(monkeypatch, hip):
    """A missing library must raise, not fall back (the product path never routes to a CPU path)."""
    monkeypatch.setattr(hip, '_lib', None)
    monkeypatch.setattr(hip, '_LIB_PATH', '/nonexistent/liblsq_hip.so')
    with pytest.raises(hip.LsqHipError):
        hip.lib()


def test_build_entry_point_checks_the_current_abi():
    """__graft_entry__.build() (what the driver runs) compiles, loads and checks the ABI version and exports."""
    import __graft_entry__
    __graft_entry__.build()
