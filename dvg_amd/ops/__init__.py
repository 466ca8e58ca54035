"""Tensor-level wrappers over the C ABI (one Python function per entry point), by kernel family.

PyTorch is plumbing here: it owns device memory and the current HIP stream; every arithmetic op of the hot path is a kernel
of libdvg_hip.so.  Activations travel as (N,C,H,W) tensors with channels_last strides, i.e. NHWC in memory.

r06: split from one 1 500-line module into `_core` (stream, timer, pointers, layouts), `conv`, `winograd`, `edge`, `dense`, `norm`,
`gp`, `backward`; `from dvg_amd import ops; ops.<name>` keeps working for every name, the module-level caches included (they
are re-exported objects, mutated in place).  Module STATE that is re-bound lives behind functions: `set_timer`,
`set_sync_bn_state` / `sync_bn_state`."""
from ._core import (  # noqa: F401
    ACT_NONE, ACT_LRELU, ACT_TANH, ACT_SIGMOID, MODE_CONV3, MODE_CONV4S2, MODE_CONVT4S2, _stream, KernelTimer,
    set_timer, _run, _p, _dev_f32, nhwc_empty, is_nhwc, to_nhwc, to_nchw, tile_policy,
)
from .conv import (  # noqa: F401
    pack_conv_weight, pack_convT_weight, packed_row_floats, pack_igemm_weight, _wp_dims, unpack_conv_weight,
    unpack_convT_weight, _splitk_ws, _SPLITK_WS, evict_captured_workspaces, _stats_buf, _tile_images, SharedBlocks,
    _MAP_CACHE, shared_map, group_sum, _check_addend, conv3x3, CONV4S2_MAX_FLOATS, conv4x4s2, convT4x4s2,
)
from .winograd import (  # noqa: F401
    winograd_weight, winograd_ok, WinoV, winograd_chain_ok, winograd_up_chain_ok, winograd_pool_chain_ok,
    conv3x3_winograd, stem_up_winograd_input,
)
from .edge import (  # noqa: F401
    conv3x3_first, first_pair_ok, conv3x3_first_pair, convT3x3_last, conv4x4s2_first, convT4x4s2_last, pixel_proj,
    _SKIP_PROJ_CACHE, clear_skip_proj_cache, _cached_skip_proj, _last_wmat, _WMAT_CACHE, _last_wmat_cached,
    precompute_skip_proj, convT_last_two_step, eval_frames, moving_mnist_compose,
)
from .dense import (  # noqa: F401
    gemm_nt, lstm_cell, lstm_cell_pre, lstm_cell_bwd, lstm_cell_x, stem_gemm, transpose2d, colsum, GEMM_TN_MAX_ROWS,
    gemm_tn, lstm_gates_bwd,
)
from .norm import (  # noqa: F401
    channel_stats, set_sync_bn_state, sync_bn_state, sync_partial_rows, bn_finalize, bn_act_apply, bn_act_bwd,
)
from .gp import (  # noqa: F401
    gp_predict, gp_var_norms, gp_trigger_step, gp_trigger_replay, gp_trigger_select, gp_elbo, gp_elbo_bwd,
    sum_steps, gp_step_group, gp_train_bwd,
)
from .backward import (  # noqa: F401
    _LOSS_W, frame_losses, mse_sum_grad, conv_wgrad_partial, conv_wgrad_partial_multi, winograd_wgrad_ok,
    winograd_wgrad_partial_multi, wgrad_finish, k4_to_w3, conv_wgrad, wgrad_thin, act_bwd, upsample2x_bwd,
)
from .._lib import check, lib  # noqa: F401
