"""torch.autograd glue for the training path (train.py:146-253).

Each fused block of dvg_amd/fused.py has a Function whose forward AND backward are kernels of
libdvg_hip.so; autograd only does the bookkeeping (which gradients are needed, when).

Gradient recipe per conv block  y = act(BN(conv(cat(up(x), skip)) + b)):
  1. dp  = (dy + maxpool-scatter(dy_pool)) * act'(y), per-channel sums       dvg_bn_act_bwd_reduce
  2. du  = A*dp + B*u + C  (batch-statistics BN backward), dgamma, dbeta     dvg_bn_bwd_finalize / dvg_affine3_apply
  3. dW  = MFMA GEMM over pixels (K split + deterministic reduce)             dvg_conv_wgrad / dvg_reduce_partials
  4. dx  = the forward implicit-GEMM kernel with re-packed weights           dvg_conv3x3_bn_act_v2 / conv4x4s2 / convT4x4s2
           (+ dvg_upsample2x_bwd when x entered through the fused nearest-x2)

Parameter gradients are written IN PLACE: the kernel that finishes a gradient (dvg_wgrad_finish, dvg_bn_bwd_finalize,
dvg_gemm_nt_bias_act / dvg_colsum with `accumulate`) adds it straight into the parameter's `.grad` buffer - a view of the
optimiser's flat gradient arena under train.Trainer - and the Function returns None for that input.  The reference's
`loss.backward()` (train.py:240) accumulates one gradient tensor per USE of a parameter (a decoder weight is used 3 S times
per iteration): that was 18 468 `at::native` add launches per 4 iterations here (5.5 % of the GPU time of a vgg_64 iteration).
autograd.DIRECT_PARAM_GRADS = False returns the gradients to autograd instead (same values; tests run both).
"""
from __future__ import annotations

import os
import weakref

import torch

from . import fused, ops
from .ops import ACT_NONE, MODE_CONV3, MODE_CONV4S2, MODE_CONVT4S2

DIRECT_PARAM_GRADS = True      # the backward kernels add parameter gradients straight into `.grad` (tests toggle it)


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


# Deferred, batched weight gradients.  A layer's weight gradient is one launch PER USE in the reference (every time step and
# every decoder call back-propagates through the same weights, train.py:213-232,240).  Our wgrad kernel splits its GEMM K
# dimension (pixels) over ~512 workgroups whose partial slabs - 75 MB per launch whatever the layer - are written and then
# reduced: 17 % of the kernel family's time.  So uses are queued per (parameter, shape) and run as ONE launch over up to
# WGRAD_BATCH of them (dvg_conv_wgrad_multi: K grows, the slabs do not), flushed when a queue is full and, through the
# autograd engine's end-of-backward callback, when the backward pass that queued them completes - `.grad` is final when
# `loss.backward()` returns, as always.  Only with in-place gradients (the finish kernel accumulates into `.grad`).
WGRAD_BATCH = 8
DENSE_BATCH = 64   # uses of a Linear / LSTMCell per batched dW GEMM (1: per use)
_wgrad_queues = {}
_wgrad_flush_queued = False


# Weight gradients of the 3x3 / stride-1 layers on maps up to 32x32 with channel counts that are multiples of 128 in
# Winograd F(4x4,3x3) form (ops.winograd_wgrad_partial_multi: a quarter of the multiply-adds; DVG_WINOGRAD_WGRAD=0: always
# the direct kernel).  Both produce the same packed slab, so the finish closures do not care.
WINOGRAD_WGRAD = os.environ.get("DVG_WINOGRAD_WGRAD", "1") != "0"


def _wino_wgrad_applies(mode, x, skip, cout, up):
    n, c, h, w = x.shape
    return bool(WINOGRAD_WGRAD and fused.WINOGRAD and mode == MODE_CONV3 and skip is None and not up
                and ops.winograd_wgrad_ok(n, c, h, w, cout))


def _wgrad_partial(mode, xs, skips, dus, up, vs=None):
    if _wino_wgrad_applies(mode, xs[0], None if skips is None else skips[0], dus[0].shape[1], up):
        return ops.winograd_wgrad_partial_multi(xs, dus, vs)
    return ops.conv_wgrad_partial_multi(mode, xs, skips, dus, upsample=up)


def _flush_one(key):
    q = _wgrad_queues.pop(key, None)
    if not q:
        return
    mode, up, finish = q[0][0], q[0][1], q[0][2]
    xs, skips, dus, vs = [e[3] for e in q], [e[4] for e in q], [e[5] for e in q], [e[6] for e in q]
    finish(_wgrad_partial(mode, xs, None if skips[0] is None else skips, dus, up, vs))


# Dense layers (Linear, LSTMCell) under BPTT: dW = sum_t dY_t^T X_t is ONE GEMM over the concatenated time steps
# (K = steps x batch) instead of two transposes, a GEMM and a column sum per step; queued like the conv weight gradients and
# flushed by the same end-of-backward callback.  Entries: (dY, [inputs...]) per use; sinks: (weight sinks..., bias sinks...).
_dense_queues = {}


def _flush_dense(key):
    ent = _dense_queues.pop(key, None)
    if not ent:
        return
    w_sinks, b_sinks, uses = ent["w"], ent["b"], ent["uses"]
    dys = [u[0] for u in uses]
    dy = dys[0] if len(dys) == 1 else torch.cat(dys, 0)
    bias = [s_b for s_b in b_sinks if s_b is not None]     # ride on the first weight GEMM (column sums of dy)
    for i, s_w in enumerate(w_sinks):
        if s_w is None:
            continue
        xs = [u[1][i] for u in uses]
        x = xs[0] if len(xs) == 1 else torch.cat(xs, 0)
        ops.gemm_tn(dy, x, out=s_w, accumulate=True, colsums=bias[:2])
        bias = bias[2:]
    for s_b in bias:
        ops.colsum(dy, out=s_b, accumulate=True)


def _dense_wgrad(w_sinks, b_sinks, dy, inputs):
    """Queue one use of a dense layer's parameter gradients (all sinks are in-place `.grad` buffers)."""
    global _wgrad_flush_queued
    key = tuple(None if t is None else t.data_ptr() for t in list(w_sinks) + list(b_sinks)) + (tuple(dy.shape),)
    ent = _dense_queues.get(key)
    if ent is None:
        ent = _dense_queues[key] = {"w": w_sinks, "b": b_sinks, "uses": []}
    ent["uses"].append((dy, inputs))
    if len(ent["uses"]) >= DENSE_BATCH:
        _flush_dense(key)
    elif len(ent["uses"]) == 1 or not _wgrad_flush_queued:   # as in _wgrad: every fresh queue asks for the flush
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrads)
        _wgrad_flush_queued = True


# Streams other than the current one on which backward nodes may have queued weight-gradient operands (train.Trainer's
# latent-path stream): the flush, which runs on the stream that called backward(), waits for them first.
JOIN_STREAMS = []


def flush_wgrads():
    global _wgrad_flush_queued
    _wgrad_flush_queued = False
    if JOIN_STREAMS and (_wgrad_queues or _dense_queues):
        cur = torch.cuda.current_stream()
        for s_ in JOIN_STREAMS:
            cur.wait_stream(s_)
    for key in list(_wgrad_queues):
        _flush_one(key)
    for key in list(_dense_queues):
        _flush_dense(key)


def drop_deferred_wgrads() -> None:
    """Forget every queued weight-gradient operand WITHOUT flushing it (rollout.drop_version_keyed_caches: after a hipGraph
    capture that raised inside backward() the queued (dY, X, V) tensors point into the freed graph pool and were never
    written; the engine skipped its end-of-backward callback, so the queues and the flag would otherwise survive)."""
    global _wgrad_flush_queued
    _wgrad_queues.clear()
    _dense_queues.clear()
    _wgrad_flush_queued = False


def _wgrad(mode, x, skip, du, up, sink, finish, tag, v=None):
    """Weight gradient of one use: queued for a batched launch when it accumulates in place into `sink`, immediate
    otherwise.  finish(partial) reduces the partial slabs into the destination.  v: the forward's Winograd input transform of
    x when it kept one (SAVE_WINO_V)."""
    global _wgrad_flush_queued
    if WGRAD_BATCH <= 1 or sink is None or not DIRECT_PARAM_GRADS:
        finish(_wgrad_partial(mode, [x], None if skip is None else [skip], [du], up, [v]))
        return
    key = (sink.data_ptr(), tag, mode, up, tuple(x.shape), tuple(du.shape), None if skip is None else tuple(skip.shape))
    q = _wgrad_queues.setdefault(key, [])
    q.append((mode, up, finish, x, skip, du, v))
    if len(q) >= WGRAD_BATCH:
        _flush_one(key)
    elif len(q) == 1 or not _wgrad_flush_queued:
        # end-of-backward flush; queued again for every fresh queue so that a pass that died half way cannot leave a
        # later pass without its flush (a second flush of the same pass finds empty queues)
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrads)
        _wgrad_flush_queued = True


def _sink(p, needed=True):
    """The buffer to accumulate parameter p's gradient into (its `.grad`, created zero-filled when missing), or None when
    in-place accumulation does not apply (switch off, not a leaf parameter, gradient not needed, odd layout)."""
    if not (DIRECT_PARAM_GRADS and needed and isinstance(p, torch.nn.Parameter) and p.requires_grad and p.is_leaf):
        return None
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    g = p.grad
    if g.dtype != torch.float32 or g.shape != p.shape or not g.is_contiguous() or g.device != p.device:
        return None
    return g


# Packed-weight cache for the training path: a parameter is re-packed once per optimizer step (its `_version`
# changes), not once per forward/backward call — train_model alone calls the encoder 2*S and the decoder 3*S times
# between two optimizer steps (train.py:213-232).
_pack_cache = {}


def _packed(weight, transposed=False, lo=None, hi=None, dim=0):
    key = (id(weight), transposed, lo, hi, dim)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
        return hit[2]
    w = weight.detach()
    if lo is not None:
        w = _c(w[lo:hi] if dim == 0 else w[:, lo:hi])
    wp = ops.pack_igemm_weight(w, transposed)
    if len(_pack_cache) > 4096:
        _pack_cache.clear()
    _pack_cache[key] = (weakref.ref(weight), (weight._version, weight.data_ptr()), wp)
    return wp


def _transposed(weight):
    """Contiguous transpose of a 2-D parameter, cached per parameter version: the data gradients of Linear / LSTMCell are
    NT GEMMs against W^T, and BPTT asked for the same transpose once per time step."""
    key = (id(weight), "T")
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
        return hit[2]
    wt = ops.transpose2d(weight.detach())
    if len(_pack_cache) > 4096:
        _pack_cache.clear()
    _pack_cache[key] = (weakref.ref(weight), (weight._version, weight.data_ptr()), wt)
    return wt


def _wino(weight, m, lo=None, hi=None, dgrad=False):
    """Winograd-domain weights U (ops.winograd_weight) of a Conv2d weight for F(m x m, 3x3), cached per parameter version:
    forward form, or - dgrad - of the flipped / transposed kernel (optionally of the input-channel slice [lo, hi)) whose
    3x3 correlation with d(out) is the data gradient."""
    key = (id(weight), "wino", m, lo, hi, dgrad)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
        return hit[2]
    w = weight.detach()
    if lo is not None:
        w = w[:, lo:hi]
    if dgrad:
        w = w.transpose(0, 1).flip(2, 3)
    u = ops.winograd_weight(w.contiguous(), m)
    if len(_pack_cache) > 4096:
        _pack_cache.clear()
    _pack_cache[key] = (weakref.ref(weight), (weight._version, weight.data_ptr()), u)
    return u


# Keep the forward's Winograd input transform V (2.25 x the layer input) for the weight gradient instead of recomputing it in
# the backward pass: ~17 GB more live memory in a vgg_64 iteration at B = 64 (of 288), half of the weight gradient's operand
# passes gone.  False: recompute.
SAVE_WINO_V = True


def _conv3_raw(x, weight, b, need_stats, lo=None, hi=None, keep_v=None):
    """Raw 3x3 conv (+ bias) of x with weight[:, lo:hi] and, for train-mode BatchNorm, its per-channel statistics: Winograd
    where fused.winograd_tile says so (statistics by one dvg_channel_stats pass over the output), else the implicit GEMM."""
    n, c, h, w = x.shape
    cout = weight.shape[0]
    m = fused.winograd_tile(n, c, h, w, cout)
    if m:
        # keep_v: a dict from a caller whose backward will take the weight gradient of this conv
        want_v = keep_v is not None and m == 4 and SAVE_WINO_V and _wino_wgrad_applies(MODE_CONV3, x, None, cout, False)
        u = ops.conv3x3_winograd(x, _wino(weight, m, lo, hi), None, b, act=ACT_NONE, return_v=want_v)
        if want_v:
            u, keep_v["v"] = u
        # (inside fused.bn_groups the statistics are per group right away: no second pass in fused.group_stats)
        return (u, ops.channel_stats(u.permute(0, 2, 3, 1).reshape(-1, cout), fused.bn_groups_now())) if need_stats else u
    wp = _packed(weight) if lo is None else _packed(weight, False, lo, hi, 1)
    return ops.conv3x3(x, None, wp, None, b, act=ACT_NONE, stats=need_stats)


def _dgrad3(du, weight, lo, hi):
    """Data gradient of a 3x3 conv w.r.t. its input channels [lo, hi): a 3x3 conv of d(out) with the flipped /
    transposed kernel - Winograd when the shape qualifies."""
    n, c, h, w = du.shape
    m = fused.winograd_tile(n, c, h, w, hi - lo)
    if m:
        return ops.conv3x3_winograd(du, _wino(weight, m, lo, hi, dgrad=True), None, None, act=ACT_NONE)
    return ops.conv3x3(du, None, _packed(weight, True, lo, hi, 1), None, None, act=ACT_NONE)


def _bn_forward(bn, u, stats, count, act, slope, pool):
    """Shared forward tail: batch (train) or running (eval) statistics -> y (, y_pool), mean, invstd.  count = elements per
    channel of the whole batch; inside fused.bn_groups(G) the statistics are per group of count / G (mean, invstd: [G][C])."""
    if bn.training:
        scale, shift, mean, invstd = fused._train_bn(bn, fused.group_stats(stats, u), count, save=True)
    else:
        with torch.no_grad():
            mean = bn.running_mean.clone()
            invstd = torch.rsqrt(bn.running_var + bn.eps)
            scale = (bn.weight.detach() * invstd).contiguous()
            shift = (bn.bias.detach() - mean * scale).contiguous()
    out = ops.bn_act_apply(u, scale, shift, act=act, slope=slope, pool=pool, inplace=False)
    return out, mean, invstd


def _upconv_weights(weight, c1):
    """K4 = W[:, :c1] (*) ones(2x2) as a ConvTranspose2d weight (C1, Cout, 4, 4) (see fused._upconv_packed: nearest-x2
    upsampling + 3x3 conv == 4x4 stride-2 transposed conv), packed for the forward (transposed mode) and for the data
    gradient (the adjoint: a plain 4x4 stride-2 conv with the same weight).  Cached per parameter version."""
    key = (id(weight), "k4", c1)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
        return hit[2]
    w = weight.detach()[:, :c1]
    k4 = torch.zeros((w.shape[0], c1, 4, 4), device=w.device, dtype=torch.float32)
    for ty in range(3):
        for tx in range(3):
            k4[:, :, 2 - ty:4 - ty, 2 - tx:4 - tx] += w[:, :, ty:ty + 1, tx:tx + 1]
    k4 = k4.permute(1, 0, 2, 3).contiguous()                       # (C1, Cout, 4, 4)
    res = (ops.pack_igemm_weight(k4, transposed=True), ops.pack_igemm_weight(k4, transposed=False))
    _pack_cache[key] = (weakref.ref(weight), (weight._version, weight.data_ptr()), res)
    return res


class _ConvBlock(torch.autograd.Function):
    """conv (+fused up/cat) + BatchNorm + activation (+ max-pool).  kinds: conv3, conv3_first, conv4s2,
    conv4s2_first, convT4s2."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias, gamma, beta, addend, cfg):
        kind, bn = cfg["kind"], cfg["bn"]
        act, slope, pool, up = cfg["act"], cfg["slope"], cfg.get("pool", False), cfg.get("upsample", False)
        b = bias.detach() if bias is not None else None
        need_stats = bn.training
        c1 = x.shape[1]
        keep = {}
        shared = cfg.get("shared")      # ops.SharedBlocks pattern: `addend` holds blocks shared by the groups of x's batch
        if shared is not None:
            addend = shared.like(addend)
        if addend is not None:
            # x half of a concat conv; `addend` = conv(skip, W_skip) shared by the decoder calls of a step (_SkipHalf)
            if kind == "conv3" and up and fused.UPCONV_AS_CONVT:
                # upsample + conv3x3 as the equivalent transposed conv: 4/9 of the MACs, forward and backward
                r = ops.convT4x4s2(x, None, _upconv_weights(weight, c1)[0], None, b, act=ACT_NONE, stats=need_stats,
                                   addend=addend)
            elif kind == "conv3":
                r = ops.conv3x3(x, None, _packed(weight, False, 0, c1, 1), None, b, upsample=up, act=ACT_NONE,
                                stats=need_stats, addend=addend)
            elif kind == "convT4s2":
                r = ops.convT4x4s2(x, None, _packed(weight, True, 0, c1, 0), None, b, act=ACT_NONE, stats=need_stats,
                                   addend=addend)
            else:
                raise RuntimeError(kind)
        elif kind == "conv3" and skip is None and not up:
            r = _conv3_raw(x, weight, b, need_stats, keep_v=keep if ctx.needs_input_grad[2] else None)
        elif kind == "conv3":
            wp = _packed(weight)
            r = ops.conv3x3(x, skip, wp, None, b, upsample=up, act=ACT_NONE, stats=need_stats)
        elif kind == "conv3_first":
            r = ops.conv3x3_first(x, weight, None, b, act=ACT_NONE, stats=need_stats)
        elif kind == "conv4s2":
            r = ops.conv4x4s2(x, _packed(weight), None, b, act=ACT_NONE, stats=need_stats)
        elif kind == "conv4s2_first":
            r = ops.conv4x4s2_first(x, weight, None, b, act=ACT_NONE, stats=need_stats)
        elif kind == "convT4s2":
            r = ops.convT4x4s2(x, skip, _packed(weight, True), None, b, act=ACT_NONE, stats=need_stats)
        else:
            raise RuntimeError(kind)
        u, st = r if need_stats else (r, None)
        n, _, h, w = u.shape
        out, mean, invstd = _bn_forward(bn, u, st, n * h * w, act, slope, pool)
        y = out[0] if pool else out
        ctx.save_for_backward(x, skip, weight, gamma, u, y, mean, invstd)
        ctx.params = (weight, bias, gamma, beta)     # the Parameter objects: backward accumulates into their .grad
        groups = mean.shape[0] if mean.dim() == 2 else 1      # time-batched: statistics (and their backward) per group
        ctx.cfg = dict(cfg, train=bn.training, count=n * h * w // groups, has_bias=bias is not None,
                       x_half=addend is not None)
        ctx.wino_v = keep.get("v")                   # the forward's Winograd input transform, for the weight gradient
        return out if pool else y

    @staticmethod
    def backward(ctx, dy, dyp=None):
        x, skip, weight, gamma, u, y, mean, invstd = ctx.saved_tensors
        cfg = ctx.cfg
        kind, act, slope, up = cfg["kind"], cfg["act"], cfg["slope"], cfg.get("upsample", False)
        dy = None if dy is None else ops.to_nhwc(dy)
        dyp = None if dyp is None else ops.to_nhwc(dyp)
        if dy is None and dyp is None:
            return (None,) * 8
        p_w, p_b, p_g, p_be = ctx.params
        ng = ctx.needs_input_grad
        # in-place sinks: BatchNorm weight / bias and the conv bias together (one finalize launch), the conv weight on its own
        s_g, s_be = _sink(p_g, ng[4]), _sink(p_be, ng[5])
        s_b = _sink(p_b, ng[3]) if cfg["has_bias"] else None
        bn_direct = s_g is not None and s_be is not None and (s_b is not None or not cfg["has_bias"])
        # x half of a shared concat conv: d(addend) = du of the sharing decoder calls is summed by the same pass that
        # produces du, into one buffer owned by the share (first call stores, later calls add)
        holder = cfg.get("ds_holder") if cfg["x_half"] and ng[6] else None
        du_sum = None
        if holder is not None:
            if holder["ds"] is None:
                holder["ds"] = torch.empty_like(u)
                du_sum = (holder["ds"], 1)
            else:
                du_sum = (holder["ds"], 2)
        du, dgamma, dbeta, dbias = ops.bn_act_bwd(dy, dyp, y, u, gamma.detach(), mean, invstd, cfg["count"], act=act,
                                                  slope=slope, train=cfg["train"],
                                                  sinks=(s_g, s_be, s_b) if bn_direct else None, du_sum=du_sum)
        if not cfg["has_bias"]:
            dbias = None
        s_w = _sink(p_w, ng[2])
        q_w = s_w                          # queue key for batched launches (None: immediate)
        beta_w = 1.0
        dW = None
        if s_w is None and ng[2]:          # gradient handed back to autograd: finished into a fresh tensor
            s_w, beta_w = torch.zeros_like(weight, memory_format=torch.contiguous_format), 0.0
            dW = s_w
        need_x, need_skip = ng[0], skip is not None and ng[1]
        dx = dskip = None
        c1 = x.shape[1]
        if cfg["x_half"]:
            # gradient of the x half only; the skip half's dgrad / wgrad happen once per step in _SkipHalf.backward,
            # which receives du (d addend = du) summed over the decoder calls that shared it.  dW lands in the
            # channel slice [0, c1) of the weight's gradient.
            if dW is not None:
                beta_w = 1.0               # fresh zero tensor: only the slice is written
            if kind == "conv3" and up and fused.UPCONV_AS_CONVT:
                if s_w is not None:
                    def fin(part, s_w=s_w, beta_w=beta_w):
                        dk4 = ops.wgrad_finish(part, torch.empty(part.shape[1:], device=part.device), 2, 4, 4)
                        # K4[.., 2-t+a, ..] += W[.., t, ..] (a = 0,1)  =>  dW[t] = sum of the 2x2 window of dK4 at (2-t), per axis
                        ops.k4_to_w3(dk4, s_w, 0, beta_w)
                    _wgrad(MODE_CONVT4S2, x, None, du, False, q_w, fin, "k4")
                if need_x:
                    dx = ops.conv4x4s2(du, _upconv_weights(weight, c1)[1], None, None, act=ACT_NONE)
            elif kind == "conv3":
                if s_w is not None:
                    _wgrad(MODE_CONV3, x, None, du, up, q_w,
                           lambda part, s_w=s_w, beta_w=beta_w, ct=weight.shape[1]: ops.wgrad_finish(
                               part, s_w, 0, 3, 3, ctot=ct, c_lo=0, beta=beta_w), "xh")
                if need_x:
                    dxu = _dgrad3(du, weight, 0, c1)
                    dx = ops.upsample2x_bwd(dxu) if up else dxu
            else:
                if s_w is not None:
                    _wgrad(MODE_CONVT4S2, x, None, du, False, q_w,
                           lambda part, s_w=s_w, beta_w=beta_w, ct=weight.shape[0]: ops.wgrad_finish(
                               part, s_w, 1, 4, 4, ctot=ct, c_lo=0, beta=beta_w), "xh")
                if need_x:
                    dx = ops.conv4x4s2(du, _packed(weight, False, 0, c1, 0), None, None, act=ACT_NONE)
            d_add = None
            if ng[6] and holder is None:
                # shared blocks (time-batched decoder calls): d(block) = sum of du over the groups that added it
                d_add = ops.group_sum(du, cfg["shared"]) if cfg.get("shared") is not None else du
            return (dx, None, dW, dbias, dgamma, dbeta, d_add, None)
        if kind == "conv3":
            if s_w is not None:
                _wgrad(MODE_CONV3, x, skip, du, up, q_w,
                       lambda part, s_w=s_w, beta_w=beta_w: ops.wgrad_finish(part, s_w, 0, 3, 3, beta=beta_w), "full",
                       v=ctx.wino_v)
                ctx.wino_v = None
            if need_x:  # dgrad = a 3x3 conv with the flipped / transposed weights (igemm or Winograd)
                dxu = _dgrad3(du, weight, 0, c1)
                dx = ops.upsample2x_bwd(dxu) if up else dxu
            if need_skip:
                dskip = _dgrad3(du, weight, c1, weight.shape[1])
        elif kind == "conv4s2":
            if s_w is not None:
                _wgrad(MODE_CONV4S2, x, None, du, False, q_w,
                       lambda part, s_w=s_w, beta_w=beta_w: ops.wgrad_finish(part, s_w, 0, 4, 4, beta=beta_w), "full")
            if need_x:
                dx = ops.convT4x4s2(du, None, _packed(weight, True), None, None, act=ACT_NONE)
        elif kind == "convT4s2":
            if s_w is not None:
                _wgrad(MODE_CONVT4S2, x, skip, du, False, q_w,
                       lambda part, s_w=s_w, beta_w=beta_w: ops.wgrad_finish(part, s_w, 1, 4, 4, beta=beta_w), "full")
            if need_x:
                dx = ops.conv4x4s2(du, _packed(weight, False, 0, c1, 0), None, None, act=ACT_NONE)
            if need_skip:
                dskip = ops.conv4x4s2(du, _packed(weight, False, c1, weight.shape[0], 0), None, None, act=ACT_NONE)
        elif kind in ("conv3_first", "conv4s2_first"):
            if need_x:
                raise RuntimeError("gradients w.r.t. the input frames are not part of the DVG training path")
            if s_w is not None:
                ops.wgrad_thin(x, du, 3 if kind == "conv3_first" else 4, out=s_w, beta=beta_w)
        else:
            raise RuntimeError(kind)
        return dx, dskip, dW, dbias, dgamma, dbeta, None, None


class _SkipHalf(torch.autograd.Function):
    """S = conv(skip, W[:, C1:]) (raw accumulators): the skip half of a decoder block's concat conv
    (vgg_64.py:98-105 / dcgan_64.py:84-86).  train_model calls the decoder three times per time step with the same skip
    tensors (train.py:227-231); with fused.share_skip_halves() the three calls share one S, so this forward and its
    dgrad / wgrad run once per step (autograd sums the three d addend = du into dS)."""

    @staticmethod
    def forward(ctx, skip, weight, cfg):
        kind, c1 = cfg["kind"], cfg["c1"]
        keep = {}
        if kind == "conv3":
            s = _conv3_raw(skip, weight, None, False, c1, weight.shape[1], keep_v=keep if ctx.needs_input_grad[1] else None)
        else:
            s = ops.convT4x4s2(skip, None, _packed(weight, True, c1, weight.shape[0], 0), None, None, act=ACT_NONE)
        ctx.save_for_backward(skip, weight)
        ctx.param = weight
        ctx.cfg = cfg
        ctx.wino_v = keep.get("v")
        ctx.set_materialize_grads(False)     # d S arrives through cfg["ds_holder"] (summed in-kernel), not through autograd
        return s

    @staticmethod
    def backward(ctx, ds):
        skip, weight = ctx.saved_tensors
        kind, c1 = ctx.cfg["kind"], ctx.cfg["c1"]
        held = ctx.cfg["ds_holder"]["ds"]
        if held is not None:
            ds = held if ds is None else ds + held
        if ds is None:
            return None, None, None
        ds = ops.to_nhwc(ds)
        s_w, dW = _sink(ctx.param, ctx.needs_input_grad[1]), None
        q_w = s_w
        if s_w is None and ctx.needs_input_grad[1]:
            s_w = dW = torch.zeros_like(weight, memory_format=torch.contiguous_format)
        dskip = None
        if kind == "conv3":     # the channel slice [c1, Ctot) of the weight's gradient
            if s_w is not None:
                _wgrad(MODE_CONV3, skip, None, ds, False, q_w,
                       lambda part, s_w=s_w, ct=weight.shape[1]: ops.wgrad_finish(part, s_w, 0, 3, 3, ctot=ct, c_lo=c1,
                                                                                  beta=1.0), "sk", v=ctx.wino_v)
                ctx.wino_v = None
            if ctx.needs_input_grad[0]:
                dskip = _dgrad3(ds, weight, c1, weight.shape[1])
        else:
            if s_w is not None:
                _wgrad(MODE_CONVT4S2, skip, None, ds, False, q_w,
                       lambda part, s_w=s_w, ct=weight.shape[0]: ops.wgrad_finish(part, s_w, 1, 4, 4, ctot=ct, c_lo=c1,
                                                                                  beta=1.0), "sk")
            if ctx.needs_input_grad[0]:
                dskip = ops.conv4x4s2(ds, _packed(weight, False, c1, weight.shape[0], 0), None, None, act=ACT_NONE)
        return dskip, dW, None


class _SplitBatch(torch.autograd.Function):
    """x (G*B, ...) -> G views of B consecutive images each (time-batched training: one encoder launch over all frames of a
    sequence, consumed frame by frame).  Backward assembles the G gradients in ONE buffer of x's layout (a slice nobody
    back-propagated into is zero-filled) - torch's own slicing would allocate and zero a full-size gradient per slice."""

    @staticmethod
    def forward(ctx, x, groups):
        b = x.shape[0] // groups
        ctx.meta = (groups, b)
        ctx.like = x          # only shape / strides / device are used
        return tuple(x[g * b:(g + 1) * b] for g in range(groups))

    @staticmethod
    def backward(ctx, *grads):
        groups, b = ctx.meta
        like = ctx.like
        if all(g is not None and g.stride() == like[:b].stride() for g in grads):
            out = torch.cat(grads, 0)
            if out.stride() == like.stride():
                return out, None
        out = torch.empty_like(like)         # preserves NHWC-in-memory strides
        for g, gr in enumerate(grads):
            if gr is None:
                out[g * b:(g + 1) * b].zero_()
            else:
                out[g * b:(g + 1) * b].copy_(gr)
        return out, None


def split_batch(x, groups):
    """G views of x's consecutive image groups; differentiable without per-slice full-size gradients (see _SplitBatch)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        b = x.shape[0] // groups
        return tuple(x[g * b:(g + 1) * b] for g in range(groups))
    return _SplitBatch.apply(x, groups)


def conv_block_autograd(kind, conv, bn, x, skip, *, upsample=False, pool=False, act, slope=0.2):
    cfg = {"kind": kind, "bn": bn, "upsample": upsample, "pool": pool, "act": act, "slope": slope}
    if kind in ("conv3", "conv4s2", "convT4s2"):
        x = ops.to_nhwc(x)
    if isinstance(skip, ops.SharedBlocks):
        # time-batched decoder calls: x holds G groups of B images, group g concatenates block skip.map[g] of the skip
        # blocks.  The skip half S = conv(skip blocks, W_skip) runs once over the DISTINCT blocks (the three calls of a
        # step, and all steps once the skip is frozen, share it - fewer than one per step); the x half adds block map[g].
        if kind not in ("conv3", "convT4s2") or pool:
            raise RuntimeError("shared skip blocks: concat blocks only")
        s = _SkipHalf.apply(skip.t, conv.weight, {"kind": kind, "c1": x.shape[1], "ds_holder": {"ds": None}})
        return _ConvBlock.apply(x, None, conv.weight, conv.bias, bn.weight, bn.bias, s, dict(cfg, shared=skip))
    share = fused.skip_share_scope()
    if share is not None and skip is not None and kind in ("conv3", "convT4s2") and not pool:
        key = (id(conv), id(skip), skip._version)
        ent = share.get(key)
        if ent is None or ent[0] is not skip:
            holder = {"ds": None}
            s = _SkipHalf.apply(ops.to_nhwc(skip), conv.weight, {"kind": kind, "c1": x.shape[1], "ds_holder": holder})
            share[key] = ent = (skip, s, holder)   # holds the skip alive while the scope lives: ids cannot be recycled
        return _ConvBlock.apply(x, None, conv.weight, conv.bias, bn.weight, bn.bias, ent[1], dict(cfg, ds_holder=ent[2]))
    return _ConvBlock.apply(x, skip, conv.weight, conv.bias, bn.weight, bn.bias, None, cfg)


class _DenseBlock(torch.autograd.Function):
    """Encoder head (Conv2d(512,dim,4,1,0)+BN+Tanh) and decoder stem (ConvTranspose2d(dim,512,4,1,0)+BN+LReLU)
    as small-M GEMMs (vgg_64.py:44-48,65-69)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, cfg):
        kind, bn, act, slope = cfg["kind"], cfg["bn"], cfg["act"], cfg["slope"]
        b = bias.detach() if bias is not None else None
        w = weight.detach()
        if kind == "head":
            n, c, h, wd = x.shape
            a = x.permute(0, 2, 3, 1).reshape(n, h * wd * c)
            gw = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()  # [dim][K]
            u2 = ops.gemm_nt(a, gw, None, b, act=ACT_NONE, splitk=max(1, min(64, a.shape[1] // 128)))
            rows, ch = n, w.shape[0]
        else:
            dim, cout, kh, kw = w.shape
            a = _c(x.reshape(-1, dim))
            n = a.shape[0]
            gw = w.permute(2, 3, 1, 0).reshape(kh * kw * cout, dim).contiguous()  # [N_out][dim]
            u2 = ops.gemm_nt(a, gw, None, b, act=ACT_NONE, period=cout).view(n * kh * kw, cout)
            rows, ch = n * kh * kw, cout
        u4 = u2.view(rows, 1, 1, ch).permute(0, 3, 1, 2)
        st = ops.channel_stats(u2, fused.bn_groups_now()) if bn.training else None
        y4, mean, invstd = _bn_forward(bn, u4, st, rows, act, slope, False)
        ctx.save_for_backward(a, gw, gamma, u4, y4, mean, invstd)
        ctx.params = (bias, gamma, beta)
        groups = mean.shape[0] if mean.dim() == 2 else 1
        ctx.cfg = dict(cfg, train=bn.training, rows=rows, ch=ch, xshape=tuple(x.shape), wshape=tuple(w.shape),
                       has_bias=bias is not None, count=rows // groups)
        if kind == "head":
            return y4.reshape(rows, ch)
        return y4.reshape(n, kh, kw, cout).permute(0, 3, 1, 2)  # NHWC-in-memory (N,512,4,4)

    @staticmethod
    def backward(ctx, dy):
        a, gw, gamma, u4, y4, mean, invstd = ctx.saved_tensors
        cfg = ctx.cfg
        rows, ch, kind = cfg["rows"], cfg["ch"], cfg["kind"]
        if kind == "head":
            dy4 = _c(dy).view(rows, 1, 1, ch).permute(0, 3, 1, 2)
        else:
            dy4 = ops.to_nhwc(dy).permute(0, 2, 3, 1).reshape(rows, 1, 1, ch).permute(0, 3, 1, 2)
        p_b, p_g, p_be = ctx.params
        ng = ctx.needs_input_grad
        s_g, s_be = _sink(p_g, ng[3]), _sink(p_be, ng[4])
        s_b = _sink(p_b, ng[2]) if cfg["has_bias"] else None
        bn_direct = s_g is not None and s_be is not None and (s_b is not None or not cfg["has_bias"])
        du4, dgamma, dbeta, dbias = ops.bn_act_bwd(dy4, None, y4, u4, gamma.detach(), mean, invstd, cfg["count"],
                                                   act=cfg["act"], slope=cfg["slope"], train=cfg["train"],
                                                   sinks=(s_g, s_be, s_b) if bn_direct else None)
        if kind == "head":
            du = du4.reshape(rows, ch)                                   # [N][dim]
            dgw = ops.gemm_tn(du, a)                                     # [dim][K]
            n, c, h, wd = cfg["xshape"]
            dW = dgw.view(ch, h, wd, c).permute(0, 3, 1, 2).contiguous()
            dx = None
            if ctx.needs_input_grad[0]:
                da = ops.gemm_nt(du, ops.transpose2d(gw), None, None)   # [N][K]
                dx = da.view(n, h, wd, c).permute(0, 3, 1, 2)
        else:
            dim, cout, kh, kw = cfg["wshape"]
            n = rows // (kh * kw)
            du = du4.reshape(n, kh * kw * cout)                          # [N][N_out]
            dgw = ops.gemm_tn(du, a)                                     # [N_out][dim]
            dW = dgw.view(kh, kw, cout, dim).permute(3, 2, 0, 1).contiguous()
            dx = None
            if ctx.needs_input_grad[0]:
                dx = ops.gemm_nt(du, ops.transpose2d(gw), None, None).view(cfg["xshape"])
        return dx, dW, (dbias if cfg["has_bias"] else None), dgamma, dbeta, None


def dense_block_autograd(kind, conv, bn, x, *, act, slope=0.2):
    cfg = {"kind": kind, "bn": bn, "act": act, "slope": slope}
    if kind == "head":
        x = ops.to_nhwc(x)
    return _DenseBlock.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, cfg)


class _LastLayer(torch.autograd.Function):
    """ConvTranspose2d(C,nc,3,1,1)+Sigmoid (vgg_64.py:88-92) / ConvTranspose2d(C,nc,4,2,1)+Tanh on cat
    (dcgan_64.py:75-79)."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias, cfg):
        nc = weight.shape[1]
        b = bias.detach() if bias is not None else None
        # the product path of the last layer (HBM-bound projection + gather), not the direct kernel (288 us at 16x3x64x64)
        ks = 3 if cfg["kind"] == "convT3" else 4
        shared = cfg.get("shared")       # time-batched decoder calls: `skip` = the distinct skip blocks
        sk = None if ks != 4 else (skip if shared is None else shared.like(skip))
        y = ops.convT_last_two_step(x, sk, weight, b, nc, ks, act=cfg["act"])
        ctx.save_for_backward(x, skip, weight, y)
        ctx.param = weight
        ctx.cfg = dict(cfg, has_bias=bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, skip, weight, y = ctx.saved_tensors
        cfg = ctx.cfg
        ks = 3 if cfg["kind"] == "convT3" else 4
        dpre = ops.act_bwd(dy, y, cfg["act"])
        first = ops.conv3x3_first if ks == 3 else ops.conv4x4s2_first
        c1 = x.shape[1]
        w = weight.detach()
        dx = dskip = None
        if ctx.needs_input_grad[0]:   # adjoint of a transposed conv = plain conv with the same weight
            dx = first(dpre, _c(w[:c1]), None, None, act=ACT_NONE)
        shared = cfg.get("shared")
        # shared skip blocks: everything on the skip side is linear in dpre, so it runs on the per-block SUM of dpre
        dpre_sk = dpre if shared is None or skip is None else ops.group_sum(dpre, shared)
        if skip is not None and ctx.needs_input_grad[1]:
            dskip = first(dpre_sk, _c(w[c1:]), None, None, act=ACT_NONE)
        s_w, dW = _sink(ctx.param, ctx.needs_input_grad[2]), None
        if s_w is not None:     # rows [0, c1) and [c1, Cin) of the ConvTranspose2d weight's gradient, in place
            ops.wgrad_thin(dpre, x, ks, out=s_w[:c1], beta=1.0)
            if skip is not None:
                ops.wgrad_thin(dpre_sk, skip, ks, out=s_w[c1:], beta=1.0)
        elif ctx.needs_input_grad[2]:
            dW = ops.wgrad_thin(dpre, x, ks)
            if skip is not None:
                dW = torch.cat([dW, ops.wgrad_thin(dpre_sk, skip, ks)], 0)
        db = dpre.sum((0, 2, 3)) if cfg["has_bias"] else None  # nc floats
        return dx, dskip, dW, db, None


def last_layer_autograd(kind, conv, x, skip, *, act):
    x = ops.to_nhwc(x)
    if isinstance(skip, ops.SharedBlocks):
        return _LastLayer.apply(x, ops.to_nhwc(skip.t), conv.weight, conv.bias, {"kind": kind, "act": act, "shared": skip})
    skip = None if skip is None else ops.to_nhwc(skip)
    return _LastLayer.apply(x, skip, conv.weight, conv.bias, {"kind": kind, "act": act})


# --------------------------------------------------------------------------------------
# GP (train mode).  Forward = dvg_gp_predict (mean, marginal variance, KL); backward = dvg_gp_train_bwd.
# --------------------------------------------------------------------------------------
def gp_train_autograd(layer, h, noise):
    from .gp_autograd import gp_train
    return gp_train(layer, h, noise)


# nn.Linear / nn.LSTMCell / the LSTM sequence: dvg_amd/autograd_recurrent.py (r06).  Imported LAST: that module reads this one's
# helpers and switches through the module object at call time.
from .autograd_recurrent import (  # noqa: E402,F401
    _Linear, _LSTMCell, _LSTMSequence, _ZERO_STATES, _zero_state, linear_autograd, lstm_cell_autograd, lstm_sequence_autograd,
)
