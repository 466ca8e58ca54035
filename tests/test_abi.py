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
    assert lib.dvg_abi_version() == 9


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
    assert h.dvg_abi_version() == 9 and h.dvg_mfma_mode() == 0 and h.dvg_packed_row_floats() == 16


def test_build_info_identifies_the_product_build():
    """dvg_build_info() (ABI 9): both shipped libraries say which arithmetic they run, carry no timing-experiment knob and
    were built from THIS tree (src = sha256 over the sources, as `make srcid` computes it); a timing-experiment knob without
    -DDVG_TIMING_EXPERIMENTS=1 does not compile (dvg_common.h), so `make all` cannot produce a wrong-results library."""
    import subprocess
    from dvg_amd import _lib
    info = _lib.build_info()
    assert info["abi"] == 9 and info["bf16x3"] == 1 and info["x3_terms"] == 6 and info["ablate"] == 0
    assert info["first_selects"] == 0 and info["timing_experiments"] == 0 and info["variant"] == "" and info["product"]
    csrc = os.path.dirname(_lib.LIB_PATH)
    srcid = subprocess.run(["make", "-s", "-C", csrc, "srcid"], capture_output=True, text=True, check=True).stdout.split()[-1]
    assert info["src"] == srcid, "libdvg_hip.so is older than its sources: rebuild (make -C dvg_amd/csrc all f32mfma)"
    h = ctypes.CDLL(os.path.join(csrc, "libdvg_hip_f32mfma.so"))
    h.dvg_build_info.restype = ctypes.c_char_p
    raw = h.dvg_build_info().decode()
    assert "bf16x3=0" in raw and "timing_experiments=0" in raw and f"src={srcid}" in raw
    common = open(os.path.join(csrc, "dvg_common.h")).read()
    assert "#error" in common and "DVG_TIMING_EXPERIMENTS" in common
    mk = open(os.path.join(csrc, "Makefile")).read()
    product_rules = mk[:mk.index("# A/B builds")]
    assert "$(DEFS)" not in product_rules, "DEFS must reach the `variant` objects only"


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
    rc = lib.dvg_gp_predict(*([one] * 7), None, None, one, None, None, None, None, 200, 90, 40, 0, 1e-3, 0, 1, None)
    assert rc == 1
    with pytest.raises(RuntimeError):
        _lib.check(rc, "gp")
    assert lib.dvg_conv_stats_rows_v2(0, 64, 64, 64, 64, 64, 0, 0) > 0
    assert lib.dvg_conv_splitk_v2(0, 64, 64, 64, 64, 64) == 1          # 2048 workgroups: no split
    assert lib.dvg_conv_splitk_v2(0, 64, 8, 8, 512, 256) == 2          # 256 workgroups, K = 32 chunks
    assert lib.dvg_conv_splitk_v2(0, 16, 8, 8, 512, 512) == 4          # per-GPU batch 16
    # r06 tile policy: LATENCY (default) = the r05 chooser, ENERGY = the 8 x 16 tile from 256 workgroups on - whose kernels never
    # split K, and the host helpers must say so too (320 workgroups would split on an 8-wide tile)
    assert lib.dvg_tile_policy() == 0
    assert lib.dvg_conv_splitk_v2(0, 10, 64, 64, 64, 64) == 1 and lib.dvg_conv_stats_rows_v2(0, 10, 64, 64, 64, 64, 0, 1) == 640
    lib.dvg_set_tile_policy(1)
    try:
        assert lib.dvg_tile_policy() == 1
        assert lib.dvg_conv_splitk_v2(0, 10, 64, 64, 64, 64) == 1
        assert lib.dvg_conv_stats_rows_v2(0, 10, 64, 64, 64, 64, 0, 1) == lib.dvg_conv_stats_rows_v2(0, 10, 64, 64, 64, 64, 0, 0) == 320
        assert lib.dvg_conv_stats_rows_v2(0, 7, 64, 64, 64, 64, 0, 1) == 448       # 224 < 256 workgroups: 8 x 8 tiles
    finally:
        lib.dvg_set_tile_policy(0)
    assert lib.dvg_tile_policy() == 0
    from dvg_amd import ops
    with ops.tile_policy(True):
        assert lib.dvg_tile_policy() == 1
        with ops.tile_policy(False):
            assert lib.dvg_tile_policy() == 0
        assert lib.dvg_tile_policy() == 1
    assert lib.dvg_tile_policy() == 0
    assert lib.dvg_gp_lds_bytes(64, 40, 1) <= 160 * 1024 and lib.dvg_gp_precision(64, 40, 1) == 64
    # r05 entry points: the checks fire before anything is launched
    assert lib.dvg_gemm_tn(one, one, None, None, None, 8, 8, 8, 8, 8, 8, 0, 0, None) == 2                  # no output (DVG_ERR_NULL)
    assert lib.dvg_gemm_tn(one, one, one, None, one, 8, 8, 8, 8, 8, 8, 0, 0, None) == 2                    # second sink without first
    assert lib.dvg_gemm_tn(one, one, one, None, None, 8, 16, 8, 8, 8, 8, 0, 0, None) == 1                  # lda < M
    assert lib.dvg_conv4x4s2_bn_act_v2(one, one, None, None, one, None, 40000, 128, 128, 64, 64, 0, 0.2, None, 0, None) == 1
    assert b"32-bit offsets" in lib.dvg_last_error()
    gp = [one] * 7 + [None, None, one, one, None, None, one]                # h .. lengthscale, noise, eps, mean, var, sample, cov, kl
    assert lib.dvg_gp_predict(*gp, 16, 990, 40, 1, 1e-3, 90, 12, None) == 1                                 # step_group > S = 11
    assert lib.dvg_gp_predict(*gp, 64, 1710, 40, 1, 1e-3, 90, 4, None) == 1 and b"fp64" in lib.dvg_last_error()   # 256 points: no fp64 fit
    # the host side's choice of steps per workgroup for a time-batched GP call (dvg_gp_step_group: B, S, latent dims, M)
    assert [lib.dvg_gp_step_group(*a) for a in ((16, 11, 90, 40), (4, 15, 90, 40), (64, 19, 90, 40), (16, 2, 90, 40),
                                               (50, 14, 90, 40), (16, 1, 90, 40))] == [6, 8, 1, 1, 1, 1]


def test_product_has_no_cpu_fallback():
    import torch
    from dvg_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv3x3_first(torch.zeros(1, 1, 8, 8), torch.zeros(64, 1, 3, 3), None, None)


def integration_snippet() -> str:
    """The fenced python block of INTEGRATION.md ("What a maintainer of the reference would add")."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(.*?)```", txt, flags=re.S)
    assert m, "INTEGRATION.md lost its binding example"
    return m.group(1)


def test_integration_document_binds_the_header_signature():
    """the example binding in INTEGRATION.md had drifted from include/dvg_hip.h (21 arguments for the
    23-argument dvg_conv3x3_bn_act_v2).  Executed here VERBATIM up to the point where it would need a GPU: the snippet
    loads the library, sets argtypes - which must equal dvg_amd/_lib.py's table, itself checked against the header - and
    defines vgg_layer_eval.  tests/test_gpu_parity.py::test_integration_snippet_runs_verbatim calls it on the GPU."""
    from dvg_amd import _lib
    ns = {}
    old = os.environ.get("DVG_HIP_LIB")
    os.environ["DVG_HIP_LIB"] = _lib.LIB_PATH
    try:
        exec(compile(integration_snippet(), "INTEGRATION.md", "exec"), ns)
    finally:
        if old is None:
            del os.environ["DVG_HIP_LIB"]
        else:
            os.environ["DVG_HIP_LIB"] = old
    h = ns["_lib"]
    for name in ("dvg_conv3x3_bn_act_v2", "dvg_pack_conv_weight_k16", "dvg_packed_row_floats"):
        restype, argtypes = _lib.SIGNATURES[name]
        fn = getattr(h, name)
        assert fn.restype is restype and list(fn.argtypes or []) == list(argtypes), name
    assert callable(ns["vgg_layer_eval"])
    # the number of arguments the document's call passes == the header's parameter count
    call = re.search(r"_lib\.dvg_conv3x3_bn_act_v2\((.*?)\)\)", integration_snippet(), flags=re.S).group(1)
    call = re.sub(r"#[^\n]*", "", call)
    n_args = len([a for a in re.sub(r"\([^()]*\)", "", call).split(",") if a.strip()])
    assert n_args == len(_lib.SIGNATURES["dvg_conv3x3_bn_act_v2"][1]) == 23, n_args


def test_python_constants_match_the_header():
    """Constants the Python shim restates from include/dvg_hip.h."""
    import re
    from dvg_amd import ops
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dvg_hip.h")).read()
    for name in ("DVG_ACT_NONE", "DVG_ACT_LRELU", "DVG_ACT_TANH", "DVG_ACT_SIGMOID"):
        v = int(re.search(rf"#define\s+{name}\s+(\d+)", text).group(1))
        assert getattr(ops, name.replace("DVG_", "")) == v, name
