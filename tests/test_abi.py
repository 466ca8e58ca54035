"""CPU: the C-ABI library loads and exports every symbol include/dvg_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dvg_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvg_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert len(syms) >= 25
    for must in ("dvg_conv3x3_bn_act_v2", "dvg_lstm_cell", "dvg_gp_predict", "dvg_gemm_nt_bias_act"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from dvg_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libdvg_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(handle, s), f"{s} declared in include/dvg_hip.h but not exported"


def test_binding_table_covers_header():
    from dvg_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.lib()
    assert lib.dvg_abi_version() == 7


def test_both_builds_of_the_library_load_and_say_which_arithmetic_they_run():
    """libdvg_hip.so forms the implicit-GEMM kernels' fp32 products as exact bf16 triples on the bf16 MFMA (packed weight
    rows of 3 x 16 bf16 = 24 floats), libdvg_hip_f32mfma.so (make f32mfma; built by __graft_entry__.build()) with the native
    f32 MFMA (rows of 16 floats); both export the same ABI."""
    from dvg_amd import _lib
    assert _lib.lib().dvg_mfma_mode() == 1 and _lib.lib().dvg_packed_row_floats() == 24
    native = os.path.join(os.path.dirname(_lib.LIB_PATH), "libdvg_hip_f32mfma.so")
    assert os.path.exists(native), "libdvg_hip_f32mfma.so not built (make -C dvg_amd/csrc f32mfma)"
    h = ctypes.CDLL(native)
    for s in header_symbols():
        assert hasattr(h, s), f"{s} missing from the f32-MFMA build"
    assert h.dvg_abi_version() == 7 and h.dvg_mfma_mode() == 0 and h.dvg_packed_row_floats() == 16


def test_host_side_checks_reject_bad_shapes_without_gpu():
    """Shape validation happens on the host before any launch, so it is testable on CPU."""
    from dvg_amd import _lib
    lib = _lib.lib()
    one = ctypes.c_void_p(16)  # fake, never dereferenced: the call must fail in the checks
    rc = lib.dvg_conv3x3_bn_act_v2(one, None, one, None, None, one, None, None, 1, 8, 8, 40, 0, 64, 0, 1, 0.2, None, 0,
                                   None, None, 0, None)
    assert rc == 1 and b"16" in lib.dvg_last_error(), lib.dvg_last_error()
    rc = lib.dvg_lstm_cell(one, one, one, one, one, one, one, one, one, None, 4, 100, None)
    assert rc == 1
    rc = lib.dvg_gp_predict(*([one] * 7), None, None, one, None, None, None, None, 200, 90, 40, 0, 1e-3, None)
    assert rc == 1
    with pytest.raises(RuntimeError):
        _lib.check(rc, "gp")
    assert lib.dvg_conv_stats_rows_v2(0, 64, 64, 64, 64, 64, 0, 0) > 0
    assert lib.dvg_conv_splitk_v2(0, 64, 64, 64, 64, 64) == 1          # 2048 workgroups: no split
    assert lib.dvg_conv_splitk_v2(0, 64, 8, 8, 512, 256) == 2          # 256 workgroups, K = 32 chunks
    assert lib.dvg_conv_splitk_v2(0, 16, 8, 8, 512, 512) == 4          # per-GPU batch 16
    assert lib.dvg_gp_lds_bytes(64, 40, 1) <= 160 * 1024 and lib.dvg_gp_precision(64, 40, 1) == 64


def test_product_has_no_cpu_fallback():
    import torch
    from dvg_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv3x3_first(torch.zeros(1, 1, 8, 8), torch.zeros(64, 1, 3, 3), None, None)
