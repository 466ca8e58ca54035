"""ctypes binding of libdvg_hip.so (the C ABI declared in include/dvg_hip.h).

The product path has NO fallback: if the shared object is missing or a symbol is
absent, `lib()` raises.  Nothing in here touches a GPU; loading works on a CPU-only
box (the HIP runtime is only initialised by the first kernel launch).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVG_HIP_LIB: an alternative BUILD of the same library (kernel A/B experiments, tools/ab_variants.sh); still no fallback
LIB_PATH = os.environ.get("DVG_HIP_LIB") or os.path.join(_HERE, "csrc", "libdvg_hip.so")

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_l = C.c_long

# name -> (restype, argtypes); must list EVERY symbol of include/dvg_hip.h
# (tests/test_abi.py parses the header and cross-checks this table).
SIGNATURES = {
    "dvg_abi_version": (_i, []),
    "dvg_last_error": (C.c_char_p, []),
    "dvg_stream_capture_id": (_l, [_p]),
    "dvg_mfma_mode": (_i, []),
    "dvg_build_info": (C.c_char_p, []),
    "dvg_set_tile_policy": (None, [_i]),
    "dvg_tile_policy": (_i, []),
    "dvg_packed_row_floats": (_i, []),
    "dvg_pack_conv_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvg_pack_convT_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvg_unpack_conv_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvg_unpack_convT_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvg_conv_first_stats_rows": (_i, [_i, _i, _i, _i]),
    "dvg_pack_conv_weight_k16": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "dvg_conv_splitk_v2": (_i, [_i, _i, _i, _i, _i, _i]),
    "dvg_conv_stats_rows_v2": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dvg_conv3x3_bn_act_v2": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _l, _p, _p, _i, _p]),
    "dvg_conv3x3_first_pair": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dvg_conv4x4s2_bn_act_v2": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _l, _p]),
    "dvg_convT4x4s2_bn_act_v2": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _l, _p, _p, _i, _p]),
    "dvg_winograd_weight": (_i, [_p, _p, _i, _i, _i, _p]),
    "dvg_winograd_input": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dvg_gemm_batched_k16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dvg_winograd_output": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p, _i, _p]),
    "dvg_winograd_output_input": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "dvg_stem_up_winograd_input": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "dvg_winograd_output_up_input": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "dvg_winograd_output_pool_input": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dvg_conv3x3_first": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "dvg_convT3x3_last": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dvg_conv4x4s2_first": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "dvg_convT4x4s2_last": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvg_moving_mnist_compose": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dvg_eval_frames": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "dvg_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _p, _p]),
    "dvg_pixel_proj": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "dvg_convT_gather": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "dvg_channel_stats_rows": (_i, [_l]),
    "dvg_channel_stats": (_i, [_p, _p, _l, _i, _i, _p]),
    "dvg_bn_finalize": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _d, _f, _f, _p, _i, _i, _p, _p]),
    "dvg_bn_running_update": (_i, [_p, _p, _i, _i, _f, _f, _f, _p, _p, _p, _i, _p]),
    "dvg_bn_act_apply": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dvg_gemm_nt_bias_act": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dvg_lstm_cell": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "dvg_lstm_cell_pre": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "dvg_lstm_cell_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "dvg_lstm_cell_x": (_i, [_p, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "dvg_stem_gemm": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "dvg_gp_precision": (_i, [_i, _i, _i]),
    "dvg_gp_bwd_precision": (_i, [_i, _i]),
    "dvg_gp_lds_bytes": (C.c_size_t, [_i, _i, _i]),
    "dvg_gp_predict": (_i, [_p] * 14 + [_i, _i, _i, _i, _f, _i, _i, _p]),
    "dvg_gp_step_group": (_i, [_i, _i, _i, _i]),
    "dvg_gp_bwd_lds_bytes": (C.c_size_t, [_i, _i]),
    "dvg_gp_bwd_chunk": (_i, [_i, _i]),
    "dvg_gemm_tn": (_i, [_p] * 5 + [_i] * 8 + [_p]),
    "dvg_gp_train_bwd": (_i, [_p] * 17 + [_i, _i, _i, _f, _i, _i, _p]),
    "dvg_sum_steps_multi": (_i, [_p, _p, _p, _i, _i, _p]),
    "dvg_gp_elbo": (_i, [_p, _p, _p, _p, _l, _l, _p, _p, _i, _i, _i, _i, _p]),
    "dvg_gp_elbo_bwd": (_i, [_p, _p, _p, _p, _l, _l, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "dvg_bn_act_bwd_rows": (_i, [_i, _i, _i, _i]),
    "dvg_bn_act_bwd_reduce": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dvg_bn_bwd_finalize": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _d, _i, _i, _i, _p]),
    "dvg_affine3_apply": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _p, _i, _i, _p]),
    "dvg_act_bwd": (_i, [_p, _p, _p, _l, _i, _f, _p]),
    "dvg_upsample2x_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvg_group_sum": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "dvg_colsum": (_i, [_p, _p, _i, _i, _i, _p]),
    "dvg_wgrad_finish": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "dvg_k4_to_w3": (_i, [_p, _p, _i, _i, _i, _i, _f, _p]),
    "dvg_reduce_partials": (_i, [_p, _p, _i, _l, _p]),
    "dvg_conv_wgrad_splits": (_i, [_i, _i, _i, _i, _i, _i]),
    "dvg_conv_wgrad": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvg_conv_wgrad_splits_multi": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "dvg_conv_wgrad_multi": (_i, [_i, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvg_winograd_wgrad_operands": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _p]),
    "dvg_winograd_wgrad_splits": (_i, [_l, _i, _i]),
    "dvg_winograd_wgrad_gemm": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "dvg_winograd_wgrad_gemm_items": (_i, [_p, _p, _i, _l, _p, _i, _i, _p]),
    "dvg_winograd_wgrad_reduce": (_i, [_p, _i, _p, _i, _i, _p]),
    "dvg_wgrad_thin_rows": (_i, [_i, _i, _i, _i]),
    "dvg_wgrad_thin": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dvg_lstm_gates_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "dvg_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvg_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvg_zero_tick": (_i, [_p, _l, _p, _p, _p, _p, _p]),
    "dvg_frame_losses_blocks": (_i, [_l]),
    "dvg_frame_losses": (_i, [_p, _p, _p, _p, _l, _i, _i, _p, _p, _p]),
    "dvg_mse_sum_grad": (_i, [_p, _p, _p, _p, _l, _f, _p]),
    "dvg_gp_var_norms": (_i, [_p, _p, _i, _i, _p]),
    "dvg_gp_trigger_step": (_i, [_p, _i, _i, _i, _p, _i, _f, _p, _p, _p, _p, _i, _p]),
    "dvg_gp_trigger_replay": (_i, [_p, _i, _p, _i, _f, _p, _p, _p]),
    "dvg_gp_trigger_select": (_i, [_p, _p, _p, _p, _i, _i, _i, _l, _p, _p, _p, _p]),
    # debug hooks (tools/diag_*.py)
    "dvg_debug_set_clockbuf": (None, [_p, C.c_uint]),
    "dvg_debug_set_gp_clockbuf": (None, [_p, C.c_uint]),
    "dvg_debug_set_wgrad_clockbuf": (None, [_p, C.c_uint]),
}

_lock = threading.Lock()
_lib = None


class DvgLibraryError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the kernel library; raise loudly when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DvgLibraryError(
                f"{LIB_PATH} not found: the HIP kernel library has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C dvg_amd/csrc`). "
                "There is no CPU/eager fallback for the DVG hot path.")
        # Two HIP runtimes live in a PyTorch-ROCm process: the one torch bundles and the system one this library links
        # (/opt/rocm).  They coexist when torch's comes up FIRST; with the library loaded before torch has initialised its
        # runtime, every launch from here fails with "no ROCm-capable device is detected" (seen with build() and smoke() in
        # one process).  So: bring torch's runtime up before the dlopen.  (No GPU - the cross-compile container, the ABI
        # tests - nothing to initialise.)
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:  # pragma: no cover - a host without torch binds the C ABI directly
            pass
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise DvgLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise DvgLibraryError(f"{LIB_PATH} lacks symbol {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        return _lib


def build_info() -> dict:
    """dvg_build_info() parsed: {"abi": 9, "bf16x3": 1, "x3_terms": 6, "ablate": 0, "first_selects": 0, "timing_experiments": 0,
    "variant": "", "src": "<sha256[:12] of the sources>"} plus "path" (the file loaded), "from_env" (DVG_HIP_LIB was set) and
    "product": True only for the library the repository ships - the default path or the in-tree f32-MFMA comparison build,
    no timing-experiment knob, no variant name."""
    raw = lib().dvg_build_info().decode()
    d = {}
    for tok in raw.split():
        k, _, v = tok.partition("=")
        d[k] = int(v) if v.lstrip("-").isdigit() else v
    d["raw"] = raw
    d["path"] = LIB_PATH
    d["from_env"] = bool(os.environ.get("DVG_HIP_LIB"))
    shipped = {os.path.join(_HERE, "csrc", "libdvg_hip.so"), os.path.join(_HERE, "csrc", "libdvg_hip_f32mfma.so")}
    d["product"] = (os.path.abspath(LIB_PATH) in shipped and d.get("x3_terms") == 6 and d.get("ablate") == 0 and
                    d.get("first_selects") == 0 and d.get("timing_experiments") == 0 and d.get("variant", "") == "")
    return d


def check(code: int, what: str = "") -> None:
    """Translate a DVG_ERR_* status into a RuntimeError (SURVEY.md §8(b) 'Errors')."""
    if code != 0:
        msg = lib().dvg_last_error()
        raise RuntimeError(f"libdvg_hip {what} failed (code {code}): {msg.decode() if msg else '?'}")
