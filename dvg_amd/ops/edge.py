"""The HBM-bound ends of the backbones (edge_layers.hip, misc_kernels.hip): first conv on the raw frame, the fused first
pair, the last transposed conv as projection + gather (with the loop-invariant skip part cached), SSIM / PSNR, Moving-MNIST
compositing."""
from __future__ import annotations

import torch

from .._lib import check, lib
from ._core import ACT_LRELU, ACT_SIGMOID, ACT_TANH, _dev_f32, _p, _run, _stream, is_nhwc, nhwc_empty
from .conv import SharedBlocks, _stats_buf, _wp_dims


def conv3x3_first(x_nchw, w, scale, shift, *, act=ACT_LRELU, slope=0.2, stats=False):
    _dev_f32(x_nchw, "conv3x3_first.x")
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    n, nc, h, wd = x.shape
    w = w.detach()
    cout = w.shape[0]
    if tuple(w.shape) != (cout, nc, 3, 3) or not w.is_contiguous():
        raise RuntimeError("conv3x3_first: weight must be contiguous (Cout,nc,3,3)")
    y = nhwc_empty(n, cout, h, wd, x.device)
    st = _stats_buf(lib().dvg_conv_first_stats_rows(3, n, h, wd), cout, x.device, 1) if stats else None
    _run("conv3x3_first", 2.0 * n * h * wd * cout * 9 * nc, 4.0 * (x.numel() + n * h * wd * cout),
         lib().dvg_conv3x3_first, _p(x), _p(w), _p(scale), _p(shift), _p(y), _p(st), n, h, wd, nc, cout, act, slope,
         _stream())
    return (y, st) if stats else y


def first_pair_ok(n, nc, h, w, cout) -> bool:
    """Shapes `conv3x3_first_pair` takes: one input channel, 8 x 16 tiles, a launch that fills the chip."""
    return nc == 1 and h % 8 == 0 and w % 16 == 0 and cout % 64 == 0 and n * (h // 8) * (w // 16) * (cout // 64) >= 512


def conv3x3_first_pair(x_nchw, w0, scale0, shift0, wp1, scale1, shift1, *, act=ACT_LRELU, slope=0.2, pool=False, y_from=0):
    """vgg_layer(1, 64) -> vgg_layer(64, Cout) (+ 2x2 max-pool) of the encoder's first stage in eval mode as ONE launch
    (dvg_conv3x3_first_pair): the 64-channel activation between the two layers is never materialised.
    w0: the first layer's (64,1,3,3) weight as a contiguous (9, 64) tensor [tap][channel].
    y_from (pool only): the full-resolution output is stored for the images [y_from, N) only - y has N - y_from images, None
    when y_from == N (a rollout discards the skip tensors of every frame but the last conditioning one)."""
    _dev_f32(x_nchw, "conv3x3_first_pair.x")
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    n, nc, h, wd = x.shape
    taps, cout, cin = _wp_dims(wp1)
    w0 = w0.detach()
    if nc != 1 or tuple(w0.shape) != (9, 64) or not w0.is_contiguous() or (taps, cin) != (9, 64) or \
            not first_pair_ok(n, nc, h, wd, cout):
        raise RuntimeError(f"conv3x3_first_pair: unsupported shapes x {tuple(x.shape)} w0 {tuple(w0.shape)} wp1 {tuple(wp1.shape)}")
    if not 0 <= y_from <= n or (y_from and not pool):
        raise RuntimeError("conv3x3_first_pair: y_from needs the pooled output and 0 <= y_from <= N")
    y = nhwc_empty(n - y_from, cout, h, wd, x.device) if y_from < n else None
    yp = nhwc_empty(n, cout, h // 2, wd // 2, x.device) if pool else None
    flops = 2.0 * n * h * wd * (64 * 9 + cout * 9 * 64)
    _run("conv3x3_igemm", flops, 4.0 * (x.numel() + (0 if y is None else y.numel()) + (0 if yp is None else yp.numel()) + wp1.numel()),
         lib().dvg_conv3x3_first_pair, _p(x), _p(w0), _p(scale0), _p(shift0), _p(wp1), _p(scale1), _p(shift1), _p(y), _p(yp),
         n, h, wd, cout, act, slope, y_from, _stream(), alg_flops=flops)
    return (y, yp) if pool else y


def convT3x3_last(x, w, bias, nc, *, act=ACT_SIGMOID):
    _dev_f32(x, "convT3x3_last.x")
    assert is_nhwc(x)
    n, cin, h, wd = x.shape
    w = w.detach()
    if tuple(w.shape) != (cin, nc, 3, 3) or not w.is_contiguous():
        raise RuntimeError("convT3x3_last: weight must be contiguous (Cin,nc,3,3)")
    y = torch.empty((n, nc, h, wd), device=x.device, dtype=torch.float32)
    _run("convT3x3_last", 2.0 * n * h * wd * cin * 9 * nc, 4.0 * (x.numel() + y.numel()),
         lib().dvg_convT3x3_last, _p(x), _p(w), _p(bias), _p(y), n, h, wd, cin, nc, act, _stream())
    return y


def conv4x4s2_first(x_nchw, w, scale, shift, *, act=ACT_LRELU, slope=0.2, stats=False):
    _dev_f32(x_nchw, "conv4x4s2_first.x")
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    n, nc, h, wd = x.shape
    w = w.detach()
    cout = w.shape[0]
    if tuple(w.shape) != (cout, nc, 4, 4) or not w.is_contiguous():
        raise RuntimeError("conv4x4s2_first: weight must be contiguous (Cout,nc,4,4)")
    y = nhwc_empty(n, cout, h // 2, wd // 2, x.device)
    st = _stats_buf(lib().dvg_conv_first_stats_rows(4, n, h, wd), cout, x.device, 1) if stats else None
    _run("conv4x4s2_first", 2.0 * n * (h // 2) * (wd // 2) * cout * 16 * nc, 4.0 * (x.numel() + y.numel()),
         lib().dvg_conv4x4s2_first, _p(x), _p(w), _p(scale), _p(shift), _p(y), _p(st), n, h, wd, nc, cout, act, slope,
         _stream())
    return (y, st) if stats else y


def convT4x4s2_last(x, skip, w, bias, nc, *, act=ACT_TANH):
    _dev_f32(x, "convT4x4s2_last.x")
    assert is_nhwc(x)
    n, c1, h, wd = x.shape
    c2 = 0
    if skip is not None:
        assert is_nhwc(skip)
        c2 = skip.shape[1]
    w = w.detach()
    if tuple(w.shape) != (c1 + c2, nc, 4, 4) or not w.is_contiguous():
        raise RuntimeError("convT4x4s2_last: weight must be contiguous (Cin,nc,4,4)")
    y = torch.empty((n, nc, 2 * h, 2 * wd), device=x.device, dtype=torch.float32)
    _run("convT4x4s2_last", 2.0 * n * h * wd * (c1 + c2) * 16 * nc,
         4.0 * (x.numel() + (skip.numel() if c2 else 0) + y.numel()), lib().dvg_convT4x4s2_last, _p(x), _p(skip),
         _p(w), _p(bias), _p(y), n, h, wd, c1, c2, nc, act, _stream())
    return y


def pixel_proj(x, wm):
    """d[px][t] = sum_c x[px][c] wm[t][c] over the pixels of an NHWC-in-memory activation (dvg_pixel_proj)."""
    assert is_nhwc(x)
    n, c, h, wd = x.shape
    t = wm.shape[0]
    d = torch.empty((n * h * wd, t), device=x.device, dtype=torch.float32)
    _run("pixel_proj", 2.0 * n * h * wd * c * t, 4.0 * (x.numel() + d.numel()), lib().dvg_pixel_proj, _p(x), _p(wm),
         _p(d), n * h * wd, c, t, _stream())
    return d


_SKIP_PROJ_CACHE = {}   # id(skip) -> (weakref(skip), skip._version, id(w), w._version, d2)


def clear_skip_proj_cache():
    """Drop cached skip projections (rollout.GraphedRollout calls this around a capture: buffers allocated while
    capturing belong to the graph's pool and must not leak into eager calls, nor the reverse)."""
    _SKIP_PROJ_CACHE.clear()


def _cached_skip_proj(skip, wm_fn, w):
    """The skip tensor of a rollout is frozen after the conditioning frames (generate_frames.py:154-157), so its
    share of the last layer's projection is computed once and reused while (tensor identity, version, weight
    version) stay the same.  Inference only (callers route training through autograd)."""
    import weakref
    key = id(skip)
    ent = _SKIP_PROJ_CACHE.get(key)
    if ent is not None and ent[0]() is skip and ent[1] == skip._version and ent[2] == id(w) and ent[3] == w._version:
        return ent[4]
    d2 = pixel_proj(skip, wm_fn())
    if len(_SKIP_PROJ_CACHE) > 8:
        _SKIP_PROJ_CACHE.clear()
    _SKIP_PROJ_CACHE[key] = (weakref.ref(skip), skip._version, id(w), w._version, d2)
    return d2


def _last_wmat(wpart, t):
    return wpart.permute(2, 3, 1, 0).reshape(t, wpart.shape[0]).contiguous()          # [(kh,kw,co)][ci]


_WMAT_CACHE = {}   # (id(w), lo, hi) -> (weakref(w), version, data_ptr, matrix)


def _last_wmat_cached(w, lo, hi, t):
    """[(kh,kw,co)][ci] projection matrix of rows lo:hi of the last layer's ConvTranspose2d weight, per weight version (it
    used to be re-permuted and copied on every decoder call)."""
    import weakref
    key = (id(w), lo, hi)
    hit = _WMAT_CACHE.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == w.data_ptr():
        return hit[3]
    m = _last_wmat(w.detach()[lo:hi], t)
    if len(_WMAT_CACHE) > 64:
        _WMAT_CACHE.clear()
    _WMAT_CACHE[key] = (weakref.ref(w), w._version, w.data_ptr(), m)
    return m


def precompute_skip_proj(skip, w, ks: int) -> None:
    """The frozen skip tensor's share of the last layer's per-pixel projection, computed ahead of the first decoder call
    (rollout.condition(), second stream); convT_last_two_step then finds it in the cache."""
    wdet = w.detach()
    c1 = wdet.shape[0] - skip.shape[1]
    t = ks * ks * wdet.shape[1]
    _cached_skip_proj(skip, lambda: _last_wmat_cached(w, c1, wdet.shape[0], t), w)


def convT_last_two_step(x, skip, w, bias, nc, ks, *, act):
    """Last layer as per-pixel projection (dvg_pixel_proj: reads the activation once) + shifted sum
    (dvg_convT_gather).  x / skip NHWC-in-memory; w the original ConvTranspose2d weight (Cin,nc,ks,ks); returns
    NCHW frames."""
    assert is_nhwc(x)
    n, c1, h, wd = x.shape
    wdet = w.detach()
    t = ks * ks * nc
    d1 = pixel_proj(x, _last_wmat_cached(w, 0, c1, t))
    d2 = None
    d2_map, d2_blk = None, 0
    if isinstance(skip, SharedBlocks):      # time-batched decoder calls: the projection of the DISTINCT skip blocks only
        d2 = pixel_proj(skip.t, _last_wmat_cached(w, c1, wdet.shape[0], t))
        d2_map, d2_blk = skip.map_dev, skip.block
        if n != skip.groups * skip.block:
            raise RuntimeError("convT_last_two_step: shared skip does not match the batch")
    elif skip is not None:
        d2 = _cached_skip_proj(skip, lambda: _last_wmat_cached(w, c1, wdet.shape[0], t), w)
    s = 2 if ks == 4 else 1
    y = torch.empty((n, nc, s * h, s * wd), device=x.device, dtype=torch.float32)
    _run("convT_gather", 0.0, 4.0 * (d1.numel() * (2 if skip is not None else 1) + y.numel()), lib().dvg_convT_gather,
         _p(d1), _p(d2), _p(bias), _p(y), ks, n, h, wd, nc, act, _p(d2_map), d2_blk, _stream())
    return y


def eval_frames(gt, pred):
    """(ssim, psnr), each (B,), of a predicted NCHW frame batch against the ground truth: per-channel skimage-style
    metrics (dvg_eval_frames) averaged over channels as utils.eval_seq does (utils.py:227-232)."""
    _dev_f32(gt, "eval_frames.gt")
    _dev_f32(pred, "eval_frames.pred")
    if gt.shape != pred.shape or gt.dim() != 4:
        raise RuntimeError(f"eval_frames: shapes {tuple(gt.shape)} vs {tuple(pred.shape)}")
    gt = gt if gt.is_contiguous() else gt.contiguous()
    pred = pred if pred.is_contiguous() else pred.contiguous()
    b, c, h, w = gt.shape
    out = torch.empty((2, b, c), device=gt.device, dtype=torch.float32)
    _run("eval_frames", 0.0, 8.0 * gt.numel(), lib().dvg_eval_frames, _p(gt), _p(pred), _p(out[0]), _p(out[1]), b * c, h,
         w, _stream())
    return out[0].mean(1), out[1].mean(1)


def moving_mnist_compose(sprites, ids, pos, seq_len, image_size):
    """(T,B,1,S,S) frames from sprites (N,D,D), ids (B,ND) int32 and pos (B,ND,T,2) int32 (dvg_moving_mnist_compose)."""
    _dev_f32(sprites, "moving_mnist_compose.sprites")
    if ids.dtype != torch.int32 or pos.dtype != torch.int32 or not ids.is_cuda or not pos.is_cuda:
        raise RuntimeError("moving_mnist_compose: ids / pos must be int32 device tensors")
    ids, pos, sprites = ids.contiguous(), pos.contiguous(), sprites.contiguous()
    b, nd = ids.shape
    if tuple(pos.shape) != (b, nd, seq_len, 2):
        raise RuntimeError(f"moving_mnist_compose: pos shape {tuple(pos.shape)}")
    out = torch.empty((seq_len, b, 1, image_size, image_size), device=sprites.device, dtype=torch.float32)
    check(lib().dvg_moving_mnist_compose(_p(sprites), _p(ids), _p(pos), _p(out), sprites.shape[0], seq_len, b, nd,
                                         image_size, sprites.shape[1], _stream()), "dvg_moving_mnist_compose")
    return out
