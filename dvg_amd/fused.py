"""Layer-level fused blocks: nn parameter containers in, HIP kernels underneath.

The model classes (dvg_amd/models/*) keep the reference's attribute tree
(nn.Conv2d / nn.BatchNorm2d / nn.LSTMCell objects as *parameter holders*, so
state_dict keys, `init_weights` and whole-module pickles stay compatible —
SURVEY.md §8(b)), but their forward never calls those modules: it calls the
functions below, which repack the weights once (cached on parameter version) and
launch the kernels of libdvg_hip.so.

Eval mode folds conv bias + BatchNorm running statistics into one per-channel
(scale, shift) pair consumed by the conv epilogue.  Train mode runs the conv with
a statistics epilogue, finalises the batch statistics on device and applies
BN + activation (+ 2x2 max-pool) in one elementwise pass.
"""
from __future__ import annotations

import os
import weakref

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_SIGMOID, ACT_TANH

_cache: "weakref.WeakKeyDictionary[nn.Module, dict]" = weakref.WeakKeyDictionary()


def _slot(mod: nn.Module) -> dict:
    d = _cache.get(mod)
    if d is None:
        d = {}
        _cache[mod] = d
    return d


def _ver(*ts) -> tuple:
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)


def packed_weight(conv: nn.Module) -> torch.Tensor:
    """[taps][Cout][Cin] repack of a Conv2d / ConvTranspose2d weight, cached per parameter version."""
    slot = _slot(conv)
    key = _ver(conv.weight)
    hit = slot.get("wp")
    if hit is not None and hit[0] == key:
        return hit[1]
    wp = ops.pack_igemm_weight(conv.weight, transposed=isinstance(conv, nn.ConvTranspose2d))
    slot["wp"] = (key, wp)
    return wp


# Winograd F(m x m, 3x3) for the eval-mode 3x3 layers (ops.conv3x3_winograd): 4x (m = 4) / 2.25x (m = 2) fewer
# multiply-adds, fp32 throughout.  Taken where it measured faster than the direct implicit GEMM at B = 64 and B = 576
# (tools/bench_winograd.py): F(4x4) on maps up to 32x32 with >= 128 input channels (x1.22-1.78 at B = 64, x1.9-2.6 at B = 576),
# F(2x2) where only its tile count fits (8x8 maps at small batch, >= 256 channels: x1.2-1.5); at 64x64 the transform passes
# (HBM-bound, 2.25-4x the activation) cost more than the GEMM saves.  Rounding: F(4x4) ~1e-5 of a layer's largest output
# (F(2x2) 2e-6, direct 3e-6), checked end to end against the oracle at the 1e-4 bar (tests/test_gpu_parity.py).
# DVG_WINOGRAD = 0: direct kernel everywhere; 2: F(2x2) only; 4 (default): both.
WINOGRAD = int(os.environ.get("DVG_WINOGRAD", "4"))
if WINOGRAD not in (0, 2, 4):
    raise RuntimeError("DVG_WINOGRAD must be 0, 2 or 4")


# F(4x4) eligibility: input channels >= 64 (the 64 -> 128 layer at 32x32 included: -1.2 % per vgg_64 rollout, alone and with
# rollouts in flight; it lost against the direct form before the chained transforms existed), maps up to 32x32 (at 64x64 the
# HBM-bound transform passes cost what the GEMM saves: 17.89 vs 17.79 ms).
_WINO_MIN_C = 64
_WINO_MAX_HW = 32


# (Cin, map side, Cout) -> tile size forced for that layer shape (0 = direct form, 2 = F(2x2)): tools/diag_layer_precision.py
# attributes the rounding of the F(4x4) transforms layer by layer with it; empty in the product
WINOGRAD_LAYER_OVERRIDE = {}


def winograd_tile(n, c, h, w, cout) -> int:
    """Winograd output-tile size (4, 2) for this eval-mode 3x3 layer, or 0 for the direct implicit GEMM."""
    if WINOGRAD_LAYER_OVERRIDE:
        forced = WINOGRAD_LAYER_OVERRIDE.get((c, h, cout))
        if forced is not None:
            return forced if (forced == 0 or ops.winograd_ok(n, c, h, w, cout, forced)) else 0
    if WINOGRAD >= 4 and c >= _WINO_MIN_C and h <= _WINO_MAX_HW and w <= _WINO_MAX_HW and ops.winograd_ok(n, c, h, w, cout, 4):
        return 4
    if WINOGRAD >= 2 and c >= 256 and ops.winograd_ok(n, c, h, w, cout, 2) and \
            ((h <= 8 and w <= 8) or (h <= 16 and w <= 16 and cout >= 256)):
        return 2
    return 0


def winograd_weight(conv: nn.Module, m: int) -> torch.Tensor:
    """U = G g G^T in the batched-GEMM layout, cached per parameter version and tile size."""
    slot = _slot(conv)
    key = _ver(conv.weight)
    hit = slot.get(("wino", m))
    if hit is not None and hit[0] == key:
        return hit[1]
    u = ops.winograd_weight(conv.weight, m)
    slot[("wino", m)] = (key, u)
    return u


# x half of a decoder block's upsample + concat conv in Winograd F(4x4) form when the skip half is hoisted (eval-mode
# rollouts); UPCONV_WINOGRAD = False: the transposed-conv (K4) form (module attribute; an environment switch until r06)
UPCONV_WINOGRAD = True
_UPCONV_WINO_MAX = 16     # largest output map side that takes this form (32 x 32 measured slower than the K4 transposed form)


def _winograd_weight_x(conv: nn.Module, c1: int) -> torch.Tensor:
    """U = G g G^T of the x half W[:, :c1] of a concat conv (F(4x4,3x3)), cached per parameter version."""
    slot = _slot(conv)
    key = (_ver(conv.weight), c1)
    hit = slot.get("wino_x")
    if hit is not None and hit[0] == key:
        return hit[1]
    u = ops.winograd_weight(conv.weight.detach()[:, :c1].contiguous(), 4)
    slot["wino_x"] = (key, u)
    return u


def gemm_weight(conv: nn.Module, kind: str) -> torch.Tensor:
    """Weights of the two dense ends as [N][K] GEMM operands in NHWC flatten order.

    kind == "head": Conv2d(512,dim,4,1,0) on a 4x4 map (vgg_64.py:44):
        W[n][ (h*4+w)*512 + c ] = w[n][c][h][w]
    kind == "stem": ConvTranspose2d(dim,512,4,1,0) on a 1x1 map (vgg_64.py:65):
        W[ (h*4+w)*512 + c ][k] = w[k][c][h][w]
    kind == "stem_t": the same transposed to [KP][N], rows zero-padded to KP in {96, 128} (dvg_stem_gemm), or None
        when dim > 128 / N % 32 != 0.
    """
    slot = _slot(conv)
    key = _ver(conv.weight)
    hit = slot.get("gw" + kind)
    if hit is not None and hit[0] == key:
        return hit[1]
    w = conv.weight.detach()
    if kind == "head":
        n, c, kh, kw = w.shape
        gw = w.permute(0, 2, 3, 1).reshape(n, kh * kw * c).contiguous()
    elif kind == "stem_t":
        k, c, kh, kw = w.shape
        gw = None
        if k <= 128 and (kh * kw * c) % 32 == 0:
            gw = torch.zeros((96 if k <= 96 else 128, kh * kw * c), device=w.device, dtype=torch.float32)
            gw[:k] = w.permute(0, 2, 3, 1).reshape(k, kh * kw * c)
    else:
        k, c, kh, kw = w.shape
        gw = w.permute(2, 3, 1, 0).reshape(kh * kw * c, k).contiguous()
    slot["gw" + kind] = (key, gw)
    return gw


def folded_affine(conv: nn.Module, bn: nn.BatchNorm2d):
    """Eval-mode BN folded with the conv bias: y = conv_nobias(x)*scale + shift."""
    slot = _slot(bn)
    key = _ver(conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    hit = slot.get("fold")
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var + bn.eps)
        scale = (bn.weight * inv) if bn.weight is not None else inv
        bias = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        shift = (bias - bn.running_mean) * scale
        if bn.bias is not None:
            shift = shift + bn.bias
        scale, shift = scale.contiguous(), shift.contiguous()
    slot["fold"] = (key, scale, shift)
    return scale, shift


def _needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(
        t is not None and (t.t if isinstance(t, ops.SharedBlocks) else t).requires_grad for t in ts)


def _no_shared(skip, who):
    if isinstance(skip, ops.SharedBlocks):
        raise RuntimeError(f"{who}: shared skip blocks (time-batched decoder calls) are a training-path operand (autograd)")


# A train-mode forward pass that stands for k identical reference passes (train.py encodes every middle frame of a
# sequence twice per closure: as step i's target and as step i+1's input - same weights, same batch, same outputs):
# k momentum updates with the same batch statistics equal ONE update with momentum 1 - (1 - m)^k, and
# num_batches_tracked advances by k.  Set by train.Trainer through `bn_passes(k)`.
_BN_PASSES = 1


class bn_passes:
    def __init__(self, k: int):
        self.k = int(k)

    def __enter__(self):
        global _BN_PASSES
        self.prev, _BN_PASSES = _BN_PASSES, self.k

    def __exit__(self, *exc):
        global _BN_PASSES
        _BN_PASSES = self.prev


def bn_momentum(bn) -> float:
    m = bn.momentum if bn.momentum is not None else 0.1
    return 1.0 - (1.0 - m) ** _BN_PASSES


def bn_passes_now() -> int:
    return _BN_PASSES


# Time-batched training (train.py:213-232 is teacher-forced: the encoder calls of a closure - and, once the latent chain has
# run, its decoder calls - are independent): G reference calls run as ONE launch over G x B images whose BatchNorm statistics
# are taken per GROUP of B consecutive images (each reference call is its own BatchNorm batch), and the running statistics
# advance group after group in the reference's call order.  `bn_groups(G, (p_first, p_mid, p_last))`: the number of groups
# and how many reference passes the first / a middle / the last group stands for (see bn_passes: a middle frame of a
# sequence is encoded twice per closure).  Inside it `bn_passes` must be 1.
_BN_GROUPS = None
_GROUP_STATS_RECOMPUTE = False      # tests set it: always take the dvg_channel_stats pass


class bn_groups:
    def __init__(self, groups: int, passes=(1, 1, 1)):
        self.val = (int(groups),) + tuple(int(k) for k in passes) if groups > 1 else None

    def __enter__(self):
        global _BN_GROUPS
        self.prev, _BN_GROUPS = _BN_GROUPS, self.val

    def __exit__(self, *exc):
        global _BN_GROUPS
        _BN_GROUPS = self.prev


def bn_groups_now() -> int:
    return _BN_GROUPS[0] if _BN_GROUPS else 1


def group_stats(st, u):
    """Partial statistics rows that respect the group boundaries: `st` itself with one group, or when its rows are per-tile
    sums in image-major order whose tiles do not straddle groups (ops._stats_buf records that: un-split igemm launches and
    the first-layer kernels), else a dvg_channel_stats pass over u (NHWC-in-memory (N,C,H,W) or [rows][C]) with per-group
    slabs.  fused._GROUP_STATS_RECOMPUTE = True always takes the extra pass (tests)."""
    g = bn_groups_now()
    if g == 1:
        return st
    if st is not None and getattr(st, "grouped", 0) == g:
        return st
    ti = getattr(st, "tile_images", 0) if st is not None else 0
    if ti and not _GROUP_STATS_RECOMPUTE and st.shape[0] % g == 0 and (u.shape[0] // g) % ti == 0:
        return st
    u2 = u if u.dim() == 2 else u.permute(0, 2, 3, 1).reshape(-1, u.shape[1])
    return ops.channel_stats(u2, g)


def bn_counter(bn):
    """The device int64 `num_batches_tracked` of a BatchNorm2d (advanced inside dvg_bn_finalize), or None."""
    return bn.num_batches_tracked if (bn.track_running_stats and bn.num_batches_tracked is not None) else None


# Recording / replaying the BatchNorm side effects of a no-grad train-mode forward: the GP fine-tuning closure encodes
# exactly what the LSTM fine-tuning closure just encoded (same frames, same encoder weights - only LSTM parameters stepped
# in between, train.py:175-198 then :146-172), so train.Trainer reuses those encodings and REPLAYS the running-statistic
# updates of the second set of passes from the recorded per-call statistics: the same dvg_bn_finalize launches on the
# same inputs in the same order - bit-identical buffers, none of the convolutions.
_BN_TRACE = None


class bn_trace:
    def __enter__(self):
        global _BN_TRACE
        self.prev, self.entries = _BN_TRACE, []
        _BN_TRACE = self.entries
        return self

    def __exit__(self, *exc):
        global _BN_TRACE
        _BN_TRACE = self.prev


def replay_bn_trace(entries) -> None:
    global _BN_TRACE, _BN_GROUPS
    prev, _BN_TRACE = _BN_TRACE, None
    prev_g = _BN_GROUPS
    try:
        for bn, stats, count, passes, groups in entries:
            _BN_GROUPS = groups
            with bn_passes(passes):
                _train_bn(bn, stats, count, synced=True)     # (recorded AFTER the cross-rank reduction: no collective here)
    finally:
        _BN_TRACE, _BN_GROUPS = prev, prev_g


# ---- synchronised BatchNorm (train.py --sync_bn; SURVEY 8(e)'s "SyncBN variant") -------------------------------------------
# The reference is ONE process whose BatchNorm statistics cover the whole batch (train.py:89-91,342-346).  Data-parallel
# training normally takes them per replica (DDP semantics, the documented deviation); with sync-BN every train-mode BatchNorm
# call sums its per-channel partial rows locally ([G][2][C]: sum and sum of squares per group), all-reduces those 2 C floats
# per group across the ranks and finalises with the GLOBAL count - N ranks x B/N clips then compute exactly what one process
# computes on B clips (tests/test_gpu_multirank.py: parameters equal after 3 iterations).  The backward pass mirrors it
# (ops.bn_act_bwd: the two per-channel sums of the BatchNorm backward are all-reduced, the parameter gradients stay local
# sums - the gradient all-reduce averages those).  Collectives cannot be captured in a hipGraph: sync-BN runs eager.
_SYNC_BN = None      # (torch.distributed module, process group, world size) or None


def set_sync_bn(dist=None, group=None) -> None:
    """Switch synchronised BatchNorm on (dist = torch.distributed with an initialised group of > 1 rank) or off (None)."""
    global _SYNC_BN
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) <= 1:
        _SYNC_BN = None
    else:
        _SYNC_BN = (dist, group, dist.get_world_size(group))
    ops.set_sync_bn_state(_SYNC_BN)


def sync_bn_world() -> int:
    return _SYNC_BN[2] if _SYNC_BN is not None else 1


def _train_bn(bn: nn.BatchNorm2d, stats, count, save=False, synced=False):
    """count = elements per channel of the WHOLE (local) batch; with groups every group has count / G of them.
    sync-BN: `stats` becomes the all-reduced sums (two fp32 rows, hi + lo, per group) and `count` the global count (`synced`: that has happened)."""
    if _SYNC_BN is not None and not synced:
        stats = ops.sync_partial_rows(stats, _BN_GROUPS[0] if _BN_GROUPS else 1)
        count = count * _SYNC_BN[2]
    if _BN_TRACE is not None:
        _BN_TRACE.append((bn, stats, count, _BN_PASSES, _BN_GROUPS))
    if _BN_GROUPS is not None:
        g, p0, p1, p2 = _BN_GROUPS
        if _BN_PASSES != 1 or count % g:
            raise RuntimeError("grouped BatchNorm: bn_passes must be 1 and the groups must divide the batch")
        m = bn.momentum if bn.momentum is not None else 0.1
        mom = tuple(1.0 - (1.0 - m) ** k for k in (p0, p1, p2))
        return ops.bn_finalize(stats, bn.weight.detach() if bn.weight is not None else None,
                               bn.bias.detach() if bn.bias is not None else None,
                               bn.running_mean if bn.track_running_stats else None,
                               bn.running_var if bn.track_running_stats else None, count // g, bn.eps, 0.0, save=save,
                               num_batches_tracked=bn_counter(bn), passes=p0 + p2 + (g - 2) * p1, groups=g,
                               group_momenta=mom)
    res = ops.bn_finalize(stats, bn.weight.detach() if bn.weight is not None else None,
                          bn.bias.detach() if bn.bias is not None else None,
                          bn.running_mean if bn.track_running_stats else None,
                          bn.running_var if bn.track_running_stats else None, count, bn.eps,
                          bn_momentum(bn), save=save, num_batches_tracked=bn_counter(bn), passes=_BN_PASSES)
    return res


# --------------------------------------------------------------------------------------
# blocks (forward).  x / skip are NHWC-in-memory; outputs likewise.
# --------------------------------------------------------------------------------------

# ---- loop-invariant skip halves -----------------------------------------------------------------------
# In a rollout the skip tensors are frozen after the conditioning frames (generate_frames.py:154-157), so in the first
# conv of every decoder block, conv(cat([up(d), skip])) = conv(up(d), W[:, :C1]) + conv(skip, W[:, C1:]), the second
# term is the same at every prediction step.  When a block sees the SAME skip tensor object (same version) for the
# second time it computes S = conv(skip, W_skip) once (raw accumulators) and from then on runs only the x half with S
# as `addend` - identical maths up to fp32 summation order, half the K loop.  A skip that changes on every call
# (training, last_frame_skip) never gets there and pays nothing.  DVG_SKIP_HOIST=0 disables it.
SKIP_HOIST = os.environ.get("DVG_SKIP_HOIST", "1") != "0"
# x half of an upsample + 3x3 conv as the equivalent 4x4 stride-2 transposed conv (see _upconv_packed); 0 disables
UPCONV_AS_CONVT = os.environ.get("DVG_UPCONV_AS_CONVT", "1") != "0"
_skip_seen = {}      # (id(conv), id(skip)) -> [weakref(skip), skip._version, weight key, sightings, S or None]


_frozen = {}         # id(skip) -> (weakref(skip), version): declared loop-invariant by the caller


def clear_skip_hoist_cache():
    _skip_seen.clear()
    _frozen.clear()


def declare_frozen_skips(skips) -> None:
    """A rollout tells the decoder blocks that these skip tensors will not change for the remaining steps
    (generate_frames.py:154-157: the skip is only refreshed while i < n_past): the first decoder call then already
    computes and uses the hoisted skip halves instead of waiting for a second sighting."""
    if not SKIP_HOIST:
        return
    for dead in [k for k, e in _frozen.items() if e[0]() is None]:
        del _frozen[dead]
    for s in skips:
        _frozen[id(s)] = (weakref.ref(s), s._version)


def _split_packed(conv, c1: int):
    """Packed weights of the x half and the skip half of a concat conv, cached per parameter version."""
    slot = _slot(conv)
    key = (_ver(conv.weight), c1)
    hit = slot.get("wp_split")
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    w = conv.weight.detach()
    tr = isinstance(conv, nn.ConvTranspose2d)
    wx, wsk = (w[:c1], w[c1:]) if tr else (w[:, :c1], w[:, c1:])
    px, ps = ops.pack_igemm_weight(wx.contiguous(), transposed=tr), ops.pack_igemm_weight(wsk.contiguous(), transposed=tr)
    slot["wp_split"] = (key, px, ps)
    return px, ps


_SHARE_SCOPE = None   # dict while inside share_skip_halves(), else None


class share_skip_halves:
    """Training-side twin of the rollout hoisting: inside this scope the decoder calls that receive the SAME skip
    tensors share one skip half per concat block, forward and backward (autograd._SkipHalf).  train.Trainer opens one
    scope per time step around the three decoder calls of train.py:227-231."""

    def __enter__(self):
        global _SHARE_SCOPE
        self.prev, _SHARE_SCOPE = _SHARE_SCOPE, {}
        return self

    def __exit__(self, *exc):
        global _SHARE_SCOPE
        _SHARE_SCOPE = self.prev


def skip_share_scope():
    return _SHARE_SCOPE


def _upconv_packed(conv, c1: int):
    """nearest-x2 upsampling followed by a 3x3 conv (pad 1) IS a stride-2 transposed conv with the 4x4 kernel
    K4 = W (*) ones(2x2): of the 9 taps of an output pixel only 4 distinct low-resolution inputs contribute.  Returns
    the packed K4 of the x half W[:, :c1] (ConvTranspose2d layout (Cin, Cout, 4, 4)), cached per parameter version:
    the x half of every decoder block's first conv (vgg_64.py:98-105) then runs on the CONVT4S2 igemm mode with 4/9 of
    the MACs.  Tap t (0..2) of the 3x3 kernel lands on k = 2 - t and k = 3 - t of the 4-tap kernel, per axis."""
    slot = _slot(conv)
    key = (_ver(conv.weight), c1)
    hit = slot.get("k4")
    if hit is not None and hit[0] == key:
        return hit[1]
    w = conv.weight.detach()[:, :c1]                      # (Cout, C1, 3, 3)
    k4 = torch.zeros((w.shape[0], c1, 4, 4), device=w.device, dtype=torch.float32)
    for ty in range(3):
        for tx in range(3):
            k4[:, :, 2 - ty:4 - ty, 2 - tx:4 - tx] += w[:, :, ty:ty + 1, tx:tx + 1]
    kp = ops.pack_igemm_weight(k4.permute(1, 0, 2, 3).contiguous(), transposed=True)
    slot["k4"] = (key, kp)
    return kp


def precompute_skip_half(conv, skip, kind: str) -> None:
    """Compute S = conv(skip, W_skip) of a decoder block NOW for a skip tensor the caller has declared frozen
    (declare_frozen_skips): rollout.condition() does this on a second stream while the LSTM warm-up runs, so that the
    first decoder call finds S ready.  kind: "conv3" (vgg blocks) | "convT4s2" (dcgan blocks)."""
    if not SKIP_HOIST or skip is None:
        return
    tr = isinstance(conv, nn.ConvTranspose2d)
    c1 = (conv.weight.shape[0] if tr else conv.weight.shape[1]) - skip.shape[1]
    _, ps = _split_packed(conv, c1)
    if kind == "conv3":
        s = ops.conv3x3(skip, None, ps, None, None, act=ACT_NONE)
    else:
        s = ops.convT4x4s2(skip, None, ps, None, None, act=ACT_NONE)
    _skip_seen[(id(conv), id(skip))] = [weakref.ref(skip), skip._version, _ver(conv.weight), 1, s]


def _hoisted_skip(conv, x, skip, partial_fn, force=False):
    """Returns (wp_x, S) when the skip half of this block is available as a precomputed addend, else None.
    force: the caller HAS to have the skip half (its x operand exists only as an upsampled WinoV, because _hoist_ready said
    yes when the producer ran): S is computed now even on what looks like a first sighting - the entry _hoist_ready saw may
    have been evicted in between (the 64-entry bound below, a dead weakref sweep)."""
    if skip is None or (not SKIP_HOIST and not force):
        return None
    k = (id(conv), id(skip))
    wkey = _ver(conv.weight)
    ent = _skip_seen.get(k)
    if ent is None or ent[0]() is not skip or ent[1] != skip._version or ent[2] != wkey:
        for dead in [kk for kk, e in _skip_seen.items() if e[0]() is None]:   # their S buffers can go
            del _skip_seen[dead]
        if len(_skip_seen) > 64:
            _skip_seen.clear()
        _skip_seen[k] = ent = [weakref.ref(skip), skip._version, wkey, 1, None]
        fz = _frozen.get(id(skip))
        if not force and (fz is None or fz[0]() is not skip or fz[1] != skip._version):
            return None                    # first sighting of an undeclared skip: the ordinary fused concat conv
    else:
        ent[3] += 1
    px, ps = _split_packed(conv, x.shape[1])
    if ent[4] is None:
        ent[4] = partial_fn(ps)            # second sighting: S = conv(skip, W_skip), raw accumulators
    return px, ent[4]


# Eval mode: two consecutive Winograd layers at 8x8 / 16x16 hand over the next layer's input transform instead of the
# activation (ops.WinoV, dvg_winograd_output_input).  DVG_WINOGRAD_CHAIN=0: every layer writes its activation.
WINOGRAD_CHAIN = os.environ.get("DVG_WINOGRAD_CHAIN", "1") != "0"


# DVG_WINOGRAD_CHAIN: 0 = every layer writes its activation; 1 = hand-overs inside a block at 8x8 / 16x16 (the r02 set);
# 2 = also at 32x32 and from the last layer of an encoder stage, through its 2x2 max-pool, to the first layer of the
# next stage (dvg_winograd_output_pool_input); 3 (default) = also from the last layer of a decoder block, through the
# nearest-x2 upsampling, to the x half of the next block's concat conv (dvg_winograd_output_up_input).
_CHAIN_LEVEL = int(os.environ.get("DVG_WINOGRAD_CHAIN", "3"))


def _chain_to(conv_next, n, c, h, w, pool=False):
    """True when the layer after this one (conv_next, fed by this layer's (n,c,h,w) output - or, `pool`, by its 2x2 max-pool -
    and by nothing else) can take a WinoV."""
    if not (WINOGRAD_CHAIN and WINOGRAD >= 4 and conv_next is not None and not conv_next.training
            and isinstance(conv_next, nn.Conv2d) and conv_next.weight.shape[1] == c and tuple(conv_next.kernel_size) == (3, 3)):
        return False
    if pool:
        return (_CHAIN_LEVEL >= 2 and winograd_tile(n, c, h // 2, w // 2, conv_next.weight.shape[0]) == 4
                and ops.winograd_pool_chain_ok(n, c, h, w))
    if h > 16 and _CHAIN_LEVEL < 2:
        return False
    return winograd_tile(n, c, h, w, conv_next.weight.shape[0]) == 4 and ops.winograd_chain_ok(n, c, h, w)


def _from(res, y_from):
    """(y, pooled) of a path that stored every image -> the `y_from` convention (y covers the images [y_from, N), None if empty)."""
    if not y_from:
        return res
    y, yp = res
    return (y[y_from:] if y_from < y.shape[0] else None), yp


def _hoist_ready(conv, skip) -> bool:
    """True when _hoisted_skip(conv, ., skip, .) will hand out the skip half on its next call (no side effects)."""
    if not SKIP_HOIST or skip is None:
        return False
    ent = _skip_seen.get((id(conv), id(skip)))
    if ent is not None and ent[0]() is skip and ent[1] == skip._version and ent[2] == _ver(conv.weight):
        return True
    fz = _frozen.get(id(skip))
    return fz is not None and fz[0]() is skip and fz[1] == skip._version


def _chain_up_to(next_up, n, c, h, w, stem=False) -> bool:
    """True when the first conv of the NEXT decoder block - next_up = (conv, bn, skip): conv(cat([up2(this output), skip])) -
    will take this layer's (n,c,h,w) output as an ops.WinoV through the upsampling: eval mode, its skip half hoisted (so the
    x half runs alone, in Winograd form), shapes dvg_winograd_output_up_input takes."""
    if next_up is None or torch.is_grad_enabled():
        return False
    conv_n, bn_n, skip_n = next_up
    if not (WINOGRAD_CHAIN and _CHAIN_LEVEL >= 3 and WINOGRAD >= 4 and UPCONV_WINOGRAD and isinstance(conv_n, nn.Conv2d)
            and tuple(conv_n.kernel_size) == (3, 3) and not conv_n.training and not bn_n.training and skip_n is not None
            and not isinstance(skip_n, ops.SharedBlocks)):
        return False
    if conv_n.weight.shape[1] - skip_n.shape[1] != c or tuple(skip_n.shape) != (n, skip_n.shape[1], 2 * h, 2 * w):
        return False
    return (2 * h <= _UPCONV_WINO_MAX and winograd_tile(n, c, 2 * h, 2 * w, conv_n.weight.shape[0]) == 4
            and ((h, w) == (4, 4) if stem else ops.winograd_up_chain_ok(n, c, h, w)) and _hoist_ready(conv_n, skip_n))


def conv3_bn_act(conv, bn, x, skip=None, *, upsample=False, pool=False, act=ACT_LRELU, slope=0.2, next_conv=None, y_from=0,
                 next_up=None):
    """vgg_layer (vgg_64.py:5-15) with optional fused cat/upsample on the input and
    fused 2x2 max-pool on the output.  next_conv: the Conv2d of the vgg_layer that consumes this layer's output and is its
    ONLY consumer - in eval mode the result may then be an ops.WinoV (that layer's Winograd input transform) instead of the
    activation; pass it on as `x` unchanged.
    y_from (pool=True only, eval mode): of the (y, pooled) pair, y - an encoder stage's skip tensor - is wanted for the images
    [y_from, N) only: the Winograd paths then do not store the rest (ops.conv3x3_winograd), the others slice.
    next_up = (conv, bn, skip) of the NEXT decoder block's first layer when this is a block's last layer and that layer is
    the only consumer: in eval mode the result may be the input transform of the upsampled output (an ops.WinoV with .up)."""
    if y_from and (not pool or bn.training or _needs_grad(x, skip, conv.weight, bn.weight)):
        raise RuntimeError("conv3_bn_act: y_from is an eval-mode, no-grad option of the pooled form")
    if isinstance(x, ops.WinoV) and x.up:
        # first conv of a decoder block fed by the previous block's last layer through the upsampling (_chain_up_to held there)
        hs = None if (bn.training or not upsample or pool) else \
            _hoisted_skip(conv, x, skip, lambda ps: ops.conv3x3(skip, None, ps, None, None, act=ACT_NONE), force=True)
        if hs is None:
            raise RuntimeError("conv3_bn_act: an upsampled WinoV needs the eval-mode concat conv with its skip half hoisted")
        sc, sh = folded_affine(conv, bn)
        n, c1, h, w = x.shape
        cout = conv.weight.shape[0]
        to_v = _chain_to(next_conv, n, cout, h, w)
        return ops.conv3x3_winograd(x, _winograd_weight_x(conv, c1), sc, sh, act=act, slope=slope, upsample=True,
                                    addend=hs[1], to_v=to_v)
    if isinstance(x, ops.WinoV):
        if bn.training or skip is not None or upsample:
            raise RuntimeError("conv3_bn_act: a WinoV input needs an eval-mode plain 3x3 layer")
        sc, sh = folded_affine(conv, bn)
        n, c, h, w = x.shape
        cout = conv.weight.shape[0]
        if winograd_tile(n, c, h, w, cout) != 4:
            raise RuntimeError("conv3_bn_act: WinoV handed to a layer that is not F(4x4,3x3)")
        to_v = _chain_to(next_conv, n, cout, h, w, pool)
        if not to_v and not pool and _chain_up_to(next_up, n, cout, h, w):
            to_v = "up"
        return ops.conv3x3_winograd(x, winograd_weight(conv, 4), sc, sh, act=act, slope=slope, pool=pool, to_v=to_v,
                                    y_from=y_from)
    if _needs_grad(x, skip, conv.weight, bn.weight):
        from .autograd import conv_block_autograd
        return conv_block_autograd("conv3", conv, bn, x, skip, upsample=upsample, pool=pool, act=act, slope=slope)
    _no_shared(skip, "conv3_bn_act")
    if not bn.training:
        sc, sh = folded_affine(conv, bn)
        if not pool:
            hs = _hoisted_skip(conv, x, skip, lambda ps: ops.conv3x3(skip, None, ps, None, None, act=ACT_NONE))
            if hs is not None:
                if upsample and UPCONV_WINOGRAD:
                    # x half of upsample + concat conv in Winograd form: the input transform reads x through the nearest-x2
                    # upsampling, the hoisted skip half enters the output transform as `addend` - a quarter of the direct
                    # form's multiplies (the K4 transposed-conv form below: 4/9), and the next layer of the block can take
                    # its input transform straight from this layer's output transform
                    n_, c1_, hx_, wx_ = x.shape
                    cout_ = conv.weight.shape[0]
                    # measured per layer pair at B = 64 (tools/_exp_up.py; this layer + the next one, K4 form -> Winograd form):
                    # 8x8 139.8 -> 125.8 us, 16x16 145.2 -> 134.4 us, 32x32 (K = 128: bandwidth-bound GEMM) 174.7 -> 189.5 us
                    if 2 * hx_ <= _UPCONV_WINO_MAX and winograd_tile(n_, c1_, 2 * hx_, 2 * wx_, cout_) == 4:
                        to_v = not torch.is_grad_enabled() and _chain_to(next_conv, n_, cout_, 2 * hx_, 2 * wx_)
                        return ops.conv3x3_winograd(x, _winograd_weight_x(conv, c1_), sc, sh, act=act, slope=slope,
                                                    upsample=True, addend=hs[1], to_v=to_v)
                if upsample and UPCONV_AS_CONVT:
                    return ops.convT4x4s2(x, None, _upconv_packed(conv, x.shape[1]), sc, sh, act=act, slope=slope,
                                          addend=hs[1])
                return ops.conv3x3(x, None, hs[0], sc, sh, upsample=upsample, act=act, slope=slope, addend=hs[1])
        if skip is None and not upsample:
            m = winograd_tile(x.shape[0], x.shape[1], x.shape[2], x.shape[3], conv.weight.shape[0])
            if m:
                to_v = m == 4 and not torch.is_grad_enabled() and \
                    _chain_to(next_conv, x.shape[0], conv.weight.shape[0], x.shape[2], x.shape[3], pool)
                if m == 4 and not to_v and not pool and _chain_up_to(next_up, x.shape[0], conv.weight.shape[0], x.shape[2], x.shape[3]):
                    to_v = "up"
                return ops.conv3x3_winograd(x, winograd_weight(conv, m), sc, sh, act=act, slope=slope, pool=pool, to_v=to_v,
                                            y_from=y_from)
        return _from(ops.conv3x3(x, skip, packed_weight(conv), sc, sh, upsample=upsample, act=act, slope=slope, pool=pool), y_from)
    wp = packed_weight(conv)
    u, st = ops.conv3x3(x, skip, wp, None, conv.bias.detach() if conv.bias is not None else None, upsample=upsample,
                        act=ACT_NONE, stats=True)
    n, _, h, w = u.shape
    sc, sh = _train_bn(bn, group_stats(st, u), n * h * w)
    return ops.bn_act_apply(u, sc, sh, act=act, slope=slope, pool=pool, inplace=True)


def conv3_first_bn_act(conv, bn, x_nchw, *, act=ACT_LRELU, slope=0.2):
    """vgg_layer(nc, 64) on the raw frame (vgg_64.py:23)."""
    if _needs_grad(x_nchw, conv.weight, bn.weight):
        from .autograd import conv_block_autograd
        return conv_block_autograd("conv3_first", conv, bn, x_nchw, None, act=act, slope=slope)
    if not bn.training:
        sc, sh = folded_affine(conv, bn)
        return ops.conv3x3_first(x_nchw, conv.weight, sc, sh, act=act, slope=slope)
    u, st = ops.conv3x3_first(x_nchw, conv.weight, None, conv.bias.detach() if conv.bias is not None else None,
                              act=ACT_NONE, stats=True)
    n, _, h, w = u.shape
    sc, sh = _train_bn(bn, group_stats(st, u), n * h * w)
    return ops.bn_act_apply(u, sc, sh, act=act, slope=slope, inplace=True)


# eval mode: the encoder's first stage vgg_layer(1, 64) -> vgg_layer(64, C) (+ pool) as one launch (FIRST_PAIR = False: two; module attribute, an environment switch until r06)
FIRST_PAIR = True


def first_pair_applies(conv0, bn0, conv1, bn1, x_nchw) -> bool:
    if not FIRST_PAIR or bn0.training or bn1.training or _needs_grad(x_nchw, conv0.weight, bn0.weight, conv1.weight, bn1.weight):
        return False
    n, nc, h, w = x_nchw.shape
    return conv0.out_channels == 64 and conv1.in_channels == 64 and ops.first_pair_ok(n, nc, h, w, conv1.out_channels)


def conv3_first_pair(conv0, bn0, conv1, bn1, x_nchw, *, pool=False, slope=0.2, y_from=0):
    """vgg_64.py:23-26 (+ :49): c1 on the raw frame, both layers in one kernel (eval mode, one input channel).
    y_from: as conv3_bn_act."""
    sc0, sh0 = folded_affine(conv0, bn0)
    sc1, sh1 = folded_affine(conv1, bn1)
    slot, key = _slot(conv0), _ver(conv0.weight)
    hit = slot.get("w_t9x64")
    if hit is None or hit[0] != key:
        hit = (key, conv0.weight.detach().reshape(64, 9).t().contiguous())      # [tap][channel]
        slot["w_t9x64"] = hit
    return ops.conv3x3_first_pair(x_nchw, hit[1], sc0, sh0, packed_weight(conv1), sc1, sh1, slope=slope, pool=pool, y_from=y_from)


def conv4s2_bn_act(conv, bn, x, *, act=ACT_LRELU, slope=0.2):
    """dcgan_conv (dcgan_64.py:4-14)."""
    if _needs_grad(x, conv.weight, bn.weight):
        from .autograd import conv_block_autograd
        return conv_block_autograd("conv4s2", conv, bn, x, None, act=act, slope=slope)
    wp = packed_weight(conv)
    if not bn.training:
        sc, sh = folded_affine(conv, bn)
        return ops.conv4x4s2(x, wp, sc, sh, act=act, slope=slope)
    u, st = ops.conv4x4s2(x, wp, None, conv.bias.detach() if conv.bias is not None else None, act=ACT_NONE,
                          stats=True)
    n, _, h, w = u.shape
    sc, sh = _train_bn(bn, group_stats(st, u), n * h * w)
    return ops.bn_act_apply(u, sc, sh, act=act, slope=slope, inplace=True)


def conv4s2_first_bn_act(conv, bn, x_nchw, *, act=ACT_LRELU, slope=0.2):
    """dcgan_conv(nc, 64) on the raw frame (dcgan_64.py:34)."""
    if _needs_grad(x_nchw, conv.weight, bn.weight):
        from .autograd import conv_block_autograd
        return conv_block_autograd("conv4s2_first", conv, bn, x_nchw, None, act=act, slope=slope)
    if not bn.training:
        sc, sh = folded_affine(conv, bn)
        return ops.conv4x4s2_first(x_nchw, conv.weight, sc, sh, act=act, slope=slope)
    u, st = ops.conv4x4s2_first(x_nchw, conv.weight, None, conv.bias.detach() if conv.bias is not None else None,
                                act=ACT_NONE, stats=True)
    n, _, h, w = u.shape
    sc, sh = _train_bn(bn, group_stats(st, u), n * h * w)
    return ops.bn_act_apply(u, sc, sh, act=act, slope=slope, inplace=True)


def convT4s2_bn_act(conv, bn, x, skip=None, *, act=ACT_LRELU, slope=0.2):
    """dcgan_upconv on cat([x, skip]) (dcgan_64.py:16-26,84-86)."""
    if _needs_grad(x, skip, conv.weight, bn.weight):
        from .autograd import conv_block_autograd
        return conv_block_autograd("convT4s2", conv, bn, x, skip, act=act, slope=slope)
    _no_shared(skip, "convT4s2_bn_act")
    if not bn.training:
        sc, sh = folded_affine(conv, bn)
        hs = _hoisted_skip(conv, x, skip, lambda ps: ops.convT4x4s2(skip, None, ps, None, None, act=ACT_NONE))
        if hs is not None:
            return ops.convT4x4s2(x, None, hs[0], sc, sh, act=act, slope=slope, addend=hs[1])
        return ops.convT4x4s2(x, skip, packed_weight(conv), sc, sh, act=act, slope=slope)
    wp = packed_weight(conv)
    u, st = ops.convT4x4s2(x, skip, wp, None, conv.bias.detach() if conv.bias is not None else None, act=ACT_NONE,
                           stats=True)
    n, _, h, w = u.shape
    sc, sh = _train_bn(bn, group_stats(st, u), n * h * w)
    return ops.bn_act_apply(u, sc, sh, act=act, slope=slope, inplace=True)


def head_bn_tanh(conv, bn, x):
    """Encoder head Conv2d(512,dim,4,1,0)+BN+Tanh on a 4x4 NHWC map (vgg_64.py:44-48): (N,dim)."""
    if _needs_grad(x, conv.weight, bn.weight):
        from .autograd import dense_block_autograd
        return dense_block_autograd("head", conv, bn, x, act=ACT_TANH)
    n, c, h, w = x.shape
    if (h, w) != tuple(conv.kernel_size):
        raise RuntimeError(f"encoder head expects a {conv.kernel_size} map, got {(h, w)}")
    gw = gemm_weight(conv, "head")
    a = x.permute(0, 2, 3, 1).reshape(n, h * w * c)  # NHWC buffer viewed as [N][K]; no copy
    k = a.shape[1]
    splitk = max(1, min(64, k // 128))
    if not bn.training:
        sc, sh = folded_affine(conv, bn)
        return ops.gemm_nt(a, gw, sc, sh, act=ACT_TANH, splitk=splitk)
    u = ops.gemm_nt(a, gw, None, conv.bias.detach() if conv.bias is not None else None, act=ACT_NONE, splitk=splitk)
    st = ops.channel_stats(u, bn_groups_now())
    sc, sh = _train_bn(bn, st, n)
    return ops.bn_act_apply(_as_nhwc_vec(u), sc, sh, act=ACT_TANH, inplace=True).reshape(n, u.shape[1])


def stem_bn_act(conv, bn, vec, *, act=ACT_LRELU, slope=0.2, next_up=None):
    """Decoder stem ConvTranspose2d(dim,512,4,1,0)+BN+LReLU (vgg_64.py:65-69): (N,dim) -> NHWC (N,512,4,4).
    next_up = (conv, bn, skip) of the first decoder block's first layer (the stem's only consumer, through `up`): in eval
    rollouts the result may be that layer's Winograd input transform instead (an ops.WinoV with .up; see _chain_up_to)."""
    if _needs_grad(vec, conv.weight, bn.weight):
        from .autograd import dense_block_autograd
        return dense_block_autograd("stem", conv, bn, vec, act=act, slope=slope)
    dim, cout, kh, kw = conv.weight.shape
    vec = vec.reshape(-1, dim)
    n = vec.shape[0]
    gw = gemm_weight(conv, "stem")
    if not bn.training:
        sc, sh = folded_affine(conv, bn)
        wt = gemm_weight(conv, "stem_t")
        if wt is not None and (kh, kw) == (4, 4) and cout % 16 == 0 and _chain_up_to(next_up, n, cout, 4, 4, stem=True):
            return ops.stem_up_winograd_input(vec, wt, dim, sc, sh, cout, act=act, slope=slope)
    out = ops.nhwc_empty(n, cout, kh, kw, vec.device)
    out2d = out.permute(0, 2, 3, 1).reshape(n, kh * kw * cout)
    if not bn.training:
        if wt is not None:
            ops.stem_gemm(vec, wt, dim, sc, sh, out2d, period=cout, act=act, slope=slope)
        else:
            ops.gemm_nt(vec, gw, sc, sh, act=act, slope=slope, period=cout, out=out2d)
        return out
    ops.gemm_nt(vec, gw, None, conv.bias.detach() if conv.bias is not None else None, act=ACT_NONE, period=cout,
                out=out2d)
    st = ops.channel_stats(out2d.view(n * kh * kw, cout), bn_groups_now())
    sc, sh = _train_bn(bn, st, n * kh * kw)
    return ops.bn_act_apply(out, sc, sh, act=act, slope=slope, inplace=True)


def _as_nhwc_vec(u2d):
    n, c = u2d.shape
    return u2d.view(n, 1, 1, c).permute(0, 3, 1, 2)


def convT3_last(conv, x, *, act=ACT_SIGMOID):
    """ConvTranspose2d(64,nc,3,1,1)+Sigmoid (vgg_64.py:88-92): NHWC in, NCHW frame out."""
    if _needs_grad(x, conv.weight):
        from .autograd import last_layer_autograd
        return last_layer_autograd("convT3", conv, x, None, act=act)
    return ops.convT_last_two_step(x, None, conv.weight, conv.bias.detach() if conv.bias is not None else None,
                                   conv.weight.shape[1], 3, act=act)


def convT4s2_last(conv, x, skip, *, act=ACT_TANH):
    """ConvTranspose2d(128,nc,4,2,1)+Tanh on cat([x,skip]) (dcgan_64.py:75-79,87)."""
    if _needs_grad(x, skip, conv.weight):
        from .autograd import last_layer_autograd
        return last_layer_autograd("convT4s2", conv, x, skip, act=act)
    _no_shared(skip, "convT4s2_last")
    return ops.convT_last_two_step(x, skip, conv.weight, conv.bias.detach() if conv.bias is not None else None,
                                   conv.weight.shape[1], 4, act=act)
