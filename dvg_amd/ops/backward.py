"""Training-only wrappers: fused frame losses, weight gradients (dense, multi-use, Winograd form), their finishing
kernels, activation / upsampling backward."""
from __future__ import annotations

import torch

from .._lib import check, lib
from ._core import MODE_CONV3, MODE_CONVT4S2, _dev_f32, _p, _run, _stream, is_nhwc, nhwc_empty


# ----------------------------------------------------------------------------------
# backward (training) wrappers
# ----------------------------------------------------------------------------------
_LOSS_W = {}     # (device, weights) -> device tensor of per-call weights


def frame_losses(pred, target, weights):
    """pred (S, K, ...) - the K decoder calls of every step -, target (S, ...) contiguous: (sums (K,), dpred like pred) with
    sums[k] = sum over steps and elements of (pred[:, k] - target)^2 and dpred = d(sum_k weights[k] sums[k]) / d pred, one pass
    (dvg_frame_losses; train.py:227-239).  `weights`: K Python floats (loss weight / elements per call)."""
    _dev_f32(pred, "frame_losses.pred")
    _dev_f32(target, "frame_losses.target")
    if not pred.is_contiguous() or not target.is_contiguous():
        raise RuntimeError("frame_losses: contiguous operands expected")
    s_, k = pred.shape[0], pred.shape[1]
    n = target[0].numel()
    if target.shape[0] != s_ or pred[0, 0].numel() != n or n % 4 or len(weights) != k or not 1 <= k <= 3:
        raise RuntimeError(f"frame_losses: pred {tuple(pred.shape)} / target {tuple(target.shape)} / {len(weights)} weights")
    key = (pred.device, tuple(float(w) for w in weights))
    w = _LOSS_W.get(key)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("frame_losses: first call with these weights during a capture (run one eager iteration first)")
        w = _LOSS_W[key] = torch.tensor(key[1], dtype=torch.float32, device=pred.device)
    sums = torch.empty(k, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    partial = torch.empty(lib().dvg_frame_losses_blocks(s_ * n) * k, device=pred.device, dtype=torch.float32)
    _run("frame_losses", 3.0 * pred.numel(), 4.0 * (2 * pred.numel() + target.numel()), lib().dvg_frame_losses, _p(pred), _p(target),
         _p(sums), _p(dpred), n, s_, k, _p(w), _p(partial), _stream())
    return sums, dpred


def mse_sum_grad(a, b, scale, need_grad=True):
    """(sum (a - b)^2 as a 0-dim tensor, 2 scale (a - b) or None) for small tensors (dvg_mse_sum_grad): a closure's latent MSE
    and its gradient (train.py:188,223)."""
    _dev_f32(a, "mse_sum_grad.a")
    _dev_f32(b, "mse_sum_grad.b")
    a, b = a.contiguous(), b.contiguous()
    if a.shape != b.shape:
        raise RuntimeError("mse_sum_grad: shape mismatch")
    out = torch.empty((), device=a.device, dtype=torch.float32)
    da = torch.empty_like(a) if need_grad else None
    check(lib().dvg_mse_sum_grad(_p(a), _p(b), _p(out), _p(da), a.numel(), float(scale), _stream()), "mse_sum_grad")
    return out, da


def conv_wgrad_partial(mode, x, skip, du, *, upsample=False):
    """The K-split partial slabs (S, taps, Cout, Cin) of a dense conv's weight gradient (dvg_conv_wgrad); finish them
    with wgrad_finish / k4_to_w3."""
    n, c1, hx, wx = x.shape
    h, w = (hx * 2, wx * 2) if upsample else (hx, wx)
    c2 = 0 if skip is None else skip.shape[1]
    cout = du.shape[1]
    cin = c1 + c2
    taps = 9 if mode == MODE_CONV3 else 16
    s = lib().dvg_conv_wgrad_splits(mode, n, h, w, cin, cout)
    if s <= 0:
        raise RuntimeError(f"conv_wgrad: unsupported shape N={n} H={h} W={w} Cin={cin} Cout={cout}")
    partial = torch.empty((s, taps, cout, cin), device=x.device, dtype=torch.float32)
    _run("conv_wgrad", 2.0 * du.numel() * taps * cin / (4 if mode == MODE_CONVT4S2 else 1),
         4.0 * (x.numel() + du.numel() + partial.numel()), lib().dvg_conv_wgrad, mode, _p(x), _p(skip), _p(du),
         _p(partial), n, h, w, c1, c2, cout, int(upsample), _stream())
    return partial


def conv_wgrad_partial_multi(mode, xs, skips, dus, *, upsample=False):
    """conv_wgrad_partial over several same-shape uses of ONE layer in one launch (dvg_conv_wgrad_multi): the partial
    slabs hold sum_i dOut_i (x) In_i.  xs / dus: lists of 1..8 tensors; skips: list or None."""
    import ctypes as C
    items = len(xs)
    x, du = xs[0], dus[0]
    n, c1, hx, wx = x.shape
    h, w = (hx * 2, wx * 2) if upsample else (hx, wx)
    c2 = 0 if skips is None else skips[0].shape[1]
    cout, cin = du.shape[1], c1 + c2
    taps = 9 if mode == MODE_CONV3 else 16
    for i in range(items):
        if xs[i].shape != x.shape or dus[i].shape != du.shape or not is_nhwc(xs[i]) or not is_nhwc(dus[i]) or \
                (c2 and (skips[i].shape != skips[0].shape or not is_nhwc(skips[i]))):
            raise RuntimeError("conv_wgrad_partial_multi: items must share one NHWC shape")
    s = lib().dvg_conv_wgrad_splits_multi(mode, n, h, w, cin, cout, items)
    if s <= 0:
        raise RuntimeError(f"conv_wgrad: unsupported shape N={n} H={h} W={w} Cin={cin} Cout={cout} items={items}")
    partial = torch.empty((s, taps, cout, cin), device=x.device, dtype=torch.float32)
    arr = C.c_void_p * items
    px, pd = arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dus])
    ps = arr(*[t.data_ptr() for t in skips]) if c2 else None
    _run("conv_wgrad", items * 2.0 * du.numel() * taps * cin / (4 if mode == MODE_CONVT4S2 else 1),
         4.0 * (items * (x.numel() + du.numel()) + partial.numel()), lib().dvg_conv_wgrad_multi, mode, items, px, ps, pd,
         _p(partial), n, h, w, c1, c2, cout, int(upsample), _stream())
    return partial


def winograd_wgrad_ok(n, cin, h, w, cout):
    """Shapes the Winograd-form weight gradient takes: 3x3 / stride 1, maps up to 32x32 in whole 4x4 tiles, channel counts
    multiples of 128 on both sides."""
    return h % 4 == 0 and w % 4 == 0 and h <= 32 and w <= 32 and cin % 128 == 0 and cout % 128 == 0 and n > 0


def winograd_wgrad_partial_multi(xs, dus, vs=None):
    """The packed (1, 9, Cout, Cin) weight-gradient slab of a 3x3 conv, sum_i dOut_i (x) In_i over 1..n same-shape uses, in
    Winograd F(4x4,3x3) form (dvg_winograd_wgrad_*): interchangeable with conv_wgrad_partial_multi(MODE_CONV3, ...).
    vs: the uses' input transforms (36, T, Cin) as the forward pass computed them (conv3x3_winograd(return_v=True)); given
    for every use (and T % 64 == 0) they are read in place instead of recomputed from xs."""
    import ctypes as C
    items = len(xs)
    x, du = xs[0], dus[0]
    n, cin, h, w = x.shape
    cout = du.shape[1]
    if not winograd_wgrad_ok(n, cin, h, w, cout) or tuple(du.shape) != (n, cout, h, w):
        raise RuntimeError(f"winograd_wgrad: unsupported shape x {tuple(x.shape)} dout {tuple(du.shape)}")
    for i in range(items):
        if xs[i].shape != x.shape or dus[i].shape != du.shape or not is_nhwc(xs[i]) or not is_nhwc(dus[i]):
            raise RuntimeError("winograd_wgrad_partial_multi: items must share one NHWC shape")
        _dev_f32(xs[i], "winograd_wgrad.x")
        _dev_f32(dus[i], "winograd_wgrad.dout")
    t = n * (h // 4) * (w // 4)
    tp = (items * t + 63) // 64 * 64
    saved = vs is not None and t % 64 == 0 and all(
        v is not None and tuple(v.shape) == (36, t, cin) and v.is_contiguous() and v.dtype == torch.float32 and v.device == x.device
        for v in vs)
    alloc = torch.empty if tp == items * t else torch.zeros      # padding rows must be zero
    v = None if saved else alloc((36, tp, cin), device=x.device, dtype=torch.float32)
    dm = alloc((36, tp, cout), device=x.device, dtype=torch.float32)
    for i in range(items):
        _run("winograd_wgrad_operands", 0.0, 4.0 * 3.25 * ((0 if saved else x.numel()) + du.numel()),
             lib().dvg_winograd_wgrad_operands, None if saved else _p(xs[i]), _p(dus[i]), _p(v), _p(dm), n, h, w, cin, cout,
             tp, i * t, _stream())
    s = lib().dvg_winograd_wgrad_splits(tp, cin, cout)
    if s <= 0:
        raise RuntimeError(f"winograd_wgrad: unsupported shape tiles={tp} Cin={cin} Cout={cout}")
    part = torch.empty((s, 36, cout, cin), device=x.device, dtype=torch.float32)
    gemm_bytes = 4.0 * (36.0 * tp * cin + dm.numel() + part.numel())
    if saved:
        pv = (C.c_void_p * items)(*[t_.data_ptr() for t_ in vs])
        _run("winograd_wgrad_gemm", 2.0 * 36 * tp * cin * cout, gemm_bytes, lib().dvg_winograd_wgrad_gemm_items, _p(dm), pv,
             items, t, _p(part), cin, cout, _stream(), alg_flops=items * 2.0 * du.numel() * 9 * cin)
    else:
        _run("winograd_wgrad_gemm", 2.0 * 36 * tp * cin * cout, gemm_bytes, lib().dvg_winograd_wgrad_gemm, _p(dm), _p(v),
             _p(part), tp, cin, cout, _stream(), alg_flops=items * 2.0 * du.numel() * 9 * cin)
    packed = torch.empty((1, 9, cout, cin), device=x.device, dtype=torch.float32)
    if s > 2:
        # many thin slabs (the 128-channel layers: one output block per position, K split ~30 ways): sum them with the wide
        # slab reduction first - the transform kernel's one thread per (co, ci) would walk 36 * S loads serially
        summed = torch.empty((36, cout, cin), device=x.device, dtype=torch.float32)
        _run("reduce_partials", 0.0, 4.0 * (part.numel() + summed.numel()), lib().dvg_reduce_partials, _p(part), _p(summed),
             s, summed.numel(), _stream())
        part, s = summed, 1
    _run("winograd_wgrad_reduce", 0.0, 4.0 * (part.numel() + packed.numel()), lib().dvg_winograd_wgrad_reduce, _p(part), s,
         _p(packed), cin, cout, _stream())
    return packed


def wgrad_finish(partial, dst, kind, kh, kw, *, ctot=None, c_lo=0, beta=0.0):
    """dst = beta * dst + sum of the partial slabs (S, kh*kw, Cout, Cin), addressed as kind 0: Conv2d weight
    (Cout, Ctot, kh, kw)[:, c_lo:c_lo+Cin]; 1: ConvTranspose2d weight (Ctot, Cout, kh, kw)[c_lo:c_lo+Cin] (flipped);
    2: plain packed (kh*kw, Cout, Cin) (dvg_wgrad_finish)."""
    s, taps, cout, cin = partial.shape
    if taps != kh * kw:
        raise RuntimeError("wgrad_finish: tap count mismatch")
    if ctot is None:
        ctot = cin
    want = (cout, ctot, kh, kw) if kind == 0 else ((ctot, cout, kh, kw) if kind == 1 else (taps, cout, cin))
    if tuple(dst.shape) != want or not dst.is_contiguous():
        raise RuntimeError(f"wgrad_finish: destination {tuple(dst.shape)} must be contiguous {want}")
    check(lib().dvg_wgrad_finish(_p(partial), s, _p(dst), kind, kh, kw, cout, cin, ctot, c_lo, float(beta), _stream()),
          "wgrad_finish")
    return dst


def k4_to_w3(dk4_packed, dw, c_lo=0, beta=0.0):
    """dw (Cout, Ctot, 3, 3)[:, c_lo:c_lo+C1] = beta * dw + 2x2 window sums of dK4 (packed (16, Cout, C1)) (dvg_k4_to_w3)."""
    t, cout, c1 = dk4_packed.shape
    if t != 16 or dw.dim() != 4 or dw.shape[0] != cout or tuple(dw.shape[2:]) != (3, 3) or not dw.is_contiguous():
        raise RuntimeError("k4_to_w3: shape mismatch")
    check(lib().dvg_k4_to_w3(_p(dk4_packed), _p(dw), cout, c1, dw.shape[1], c_lo, float(beta), _stream()), "k4_to_w3")
    return dw


def conv_wgrad(mode, x, skip, du, *, upsample=False):
    """Packed weight gradient [taps][Cout][Cin] of a dense conv (see dvg_conv_wgrad)."""
    n, c1, hx, wx = x.shape
    h, w = (hx * 2, wx * 2) if upsample else (hx, wx)
    c2 = 0 if skip is None else skip.shape[1]
    cout = du.shape[1]
    cin = c1 + c2
    taps = 9 if mode == MODE_CONV3 else 16
    s = lib().dvg_conv_wgrad_splits(mode, n, h, w, cin, cout)
    if s <= 0:
        raise RuntimeError(f"conv_wgrad: unsupported shape N={n} H={h} W={w} Cin={cin} Cout={cout}")
    partial = torch.empty((s, taps, cout, cin), device=x.device, dtype=torch.float32)
    _run("conv_wgrad", 2.0 * du.numel() * taps * cin / (4 if mode == MODE_CONVT4S2 else 1),
         4.0 * (x.numel() + du.numel() + partial.numel()), lib().dvg_conv_wgrad, mode, _p(x), _p(skip), _p(du),
         _p(partial), n, h, w, c1, c2, cout, int(upsample), _stream())
    if s == 1:
        return partial[0]
    out = torch.empty((taps, cout, cin), device=x.device, dtype=torch.float32)
    check(lib().dvg_reduce_partials(_p(partial), _p(out), s, out.numel(), _stream()), "reduce_partials")
    return out


def wgrad_thin(inp_nchw, dout_nhwc, ks, out=None, beta=0.0):
    """dW (C, nc, ks, ks) of a thin layer (see dvg_wgrad_thin); `out` (contiguous, same shape): out = beta * out + dW."""
    inp = inp_nchw if inp_nchw.is_contiguous() else inp_nchw.contiguous()
    n, nc, hi, wi = inp.shape
    c = dout_nhwc.shape[1]
    assert is_nhwc(dout_nhwc)
    rows = lib().dvg_wgrad_thin_rows(ks, n, hi, wi)
    partial = torch.empty((rows, c, nc * ks * ks), device=inp.device, dtype=torch.float32)
    _run("wgrad_thin", 2.0 * dout_nhwc.numel() * nc * ks * ks, 4.0 * (inp.numel() + dout_nhwc.numel()),
         lib().dvg_wgrad_thin, _p(inp), _p(dout_nhwc), _p(partial), ks, n, hi, wi, nc, c, _stream())
    if out is None:
        out = torch.empty((c, nc, ks, ks), device=inp.device, dtype=torch.float32)
        beta = 0.0
    elif tuple(out.shape) != (c, nc, ks, ks) or not out.is_contiguous():
        raise RuntimeError("wgrad_thin: bad destination")
    n = out.numel()
    check(lib().dvg_wgrad_finish(_p(partial), rows, _p(out), 2, 1, 1, 1, n, n, 0, float(beta), _stream()),
          "wgrad_finish(thin)")
    return out


def act_bwd(dy, y, act, slope=0.0):
    dy = dy if dy.is_contiguous() else dy.contiguous()
    y = y if y.is_contiguous() else y.contiguous()
    out = torch.empty_like(y)
    check(lib().dvg_act_bwd(_p(dy), _p(y), _p(out), y.numel(), act, slope, _stream()), "act_bwd")
    return out


def upsample2x_bwd(dxu):
    assert is_nhwc(dxu)
    n, c, h2, w2 = dxu.shape
    dx = nhwc_empty(n, c, h2 // 2, w2 // 2, dxu.device)
    check(lib().dvg_upsample2x_bwd(_p(dxu), _p(dx), n, h2 // 2, w2 // 2, c, _stream()), "upsample2x_bwd")
    return dx
