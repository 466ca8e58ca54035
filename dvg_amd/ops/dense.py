"""Dense / recurrent wrappers (dense.hip): NT / TN GEMMs with fused bias + activation, the LSTM cell kernels, the decoder
stem, transposes and column sums."""
from __future__ import annotations

import torch

from .._lib import check, lib
from ._core import ACT_LRELU, ACT_NONE, _dev_f32, _p, _run, _stream


# ----------------------------------------------------------------------------------
# dense / recurrent
# ----------------------------------------------------------------------------------
def gemm_nt(a, w, scale, shift, *, act=ACT_NONE, slope=0.0, period=None, splitk=1, out=None, accumulate=False):
    """out[m][n] = act((sum_k a[m][k] w[n][k]) * scale[n%period] + shift[n%period]); accumulate: out += (needs `out`)."""
    _dev_f32(a, "gemm_nt.a")
    _dev_f32(w, "gemm_nt.w")
    if a.dim() != 2 or a.stride(1) != 1:
        a = a.contiguous().view(a.shape[0], -1)
    w = w if w.is_contiguous() else w.contiguous()
    m, k = a.shape
    n, kw = w.shape
    if kw != k:
        raise RuntimeError(f"gemm_nt: K mismatch {k} vs {kw}")
    if period is None:
        period = n
    if out is None:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    ws = torch.empty((splitk, m, n), device=a.device, dtype=torch.float32) if splitk > 1 else None
    _run("gemm_nt", 2.0 * m * n * k, 4.0 * (m * k + n * k + m * n), lib().dvg_gemm_nt_bias_act, _p(a), _p(w),
         _p(scale), _p(shift), _p(out), _p(ws), m, n, k, a.stride(0), out.stride(0), period, splitk, act, slope,
         int(accumulate), _stream())
    return out


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh, want_gates=False):
    for t, nm in ((x, "x"), (h, "h"), (c, "c")):
        _dev_f32(t, "lstm_cell." + nm)
    b, hid = h.shape
    x = x if x.is_contiguous() else x.contiguous()
    h = h if h.is_contiguous() else h.contiguous()
    c = c if c.is_contiguous() else c.contiguous()
    if tuple(x.shape) != (b, hid) or tuple(w_ih.shape) != (4 * hid, hid) or tuple(w_hh.shape) != (4 * hid, hid):
        raise RuntimeError("lstm_cell: shape mismatch (input size must equal hidden size)")
    h_out = torch.empty_like(h)
    c_out = torch.empty_like(c)
    gates = torch.empty((b, 4 * hid), device=h.device, dtype=torch.float32) if want_gates else None
    _run("lstm_cell", 2.0 * b * 4 * hid * 2 * hid, 4.0 * (8 * hid * hid + 5 * b * hid), lib().dvg_lstm_cell, _p(x),
         _p(h), _p(c), _p(w_ih.detach()), _p(w_hh.detach()), _p(b_ih.detach()), _p(b_hh.detach()), _p(h_out),
         _p(c_out), _p(gates), b, hid, _stream())
    return (h_out, c_out, gates) if want_gates else (h_out, c_out)


def lstm_cell_pre(pre, h, c, w_hh, h_out, c_out, gates_out):
    """One LSTMCell step whose input half `pre` = W_ih x + b_ih + b_hh (B,4H) is given (dvg_lstm_cell_pre); writes into the
    caller's h_out / c_out / gates_out (slices of the per-sequence buffers of autograd._LSTMSequence)."""
    b, hid = h.shape
    _run("lstm_cell", 2.0 * b * 4 * hid * hid, 4.0 * (4 * hid * hid + 9 * b * hid), lib().dvg_lstm_cell_pre, _p(pre), _p(h),
         _p(c), _p(w_hh), _p(h_out), _p(c_out), _p(gates_out), b, hid, _stream())


def lstm_cell_bwd(dh_a, dh_b, dc, gates, c_prev, c_new, w_hh_t, dG, dc_prev, dh_prev):
    """One BPTT step of an LSTMCell in one launch (dvg_lstm_cell_bwd): dG, dc_prev and dh_prev = dG W_hh."""
    b, hid = c_new.shape
    _run("lstm_cell_bwd", 2.0 * b * 4 * hid * hid, 4.0 * (4 * hid * hid + 12 * b * hid), lib().dvg_lstm_cell_bwd, _p(dh_a),
         _p(dh_b), _p(dc), _p(gates), _p(c_prev), _p(c_new), _p(w_hh_t), _p(dG), _p(dc_prev), _p(dh_prev), b, hid, _stream())


def lstm_cell_x(x, h, c, w_x, w_hh, bias):
    """First cell of a time step with the embedding folded in (dvg_lstm_cell_x): x (B,Kx) raw LSTM input, w_x = W_ih W_e
    zero-padded to (4H,Kxp), bias = W_ih b_e + b_ih + b_hh.  Inference path."""
    for t, nm in ((x, "x"), (h, "h"), (c, "c")):
        _dev_f32(t, "lstm_cell_x." + nm)
    b, hid = h.shape
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.contiguous().view(x.shape[0], -1)
    h = h if h.is_contiguous() else h.contiguous()
    c = c if c.is_contiguous() else c.contiguous()
    kx = x.shape[1]
    if x.shape[0] != b or tuple(w_x.shape) != (4 * hid, w_x.shape[1]) or w_x.shape[1] < kx or \
            tuple(w_hh.shape) != (4 * hid, hid) or bias.numel() != 4 * hid:
        raise RuntimeError("lstm_cell_x: shape mismatch")
    h_out, c_out = torch.empty_like(h), torch.empty_like(c)
    _run("lstm_cell", 2.0 * b * 4 * hid * (kx + hid), 4.0 * (4 * hid * (kx + hid) + 5 * b * hid), lib().dvg_lstm_cell_x,
         _p(x), x.stride(0), kx, _p(h), _p(c), _p(w_x), w_x.shape[1], _p(w_hh.detach()), _p(bias), _p(h_out), _p(c_out),
         b, hid, _stream())
    return h_out, c_out


def stem_gemm(vec, w_kn, k, scale, shift, out, *, period, act=ACT_LRELU, slope=0.2):
    """out[m][n] = act((vec[m][:k] . w_kn[:k][n]) * scale[n % period] + shift[n % period]) (dvg_stem_gemm); w_kn is the
    zero-padded transposed GEMM weight (KP,N), KP in {96, 128}."""
    _dev_f32(vec, "stem_gemm.vec")
    if vec.dim() != 2 or vec.stride(1) != 1:
        vec = vec.contiguous().view(vec.shape[0], -1)
    m = vec.shape[0]
    kp, n = w_kn.shape
    if vec.shape[1] != k or tuple(out.shape) != (m, n) or out.stride(1) != 1:
        raise RuntimeError("stem_gemm: shape mismatch")
    _run("gemm_nt", 2.0 * m * n * k, 4.0 * (m * k + n * k + m * n), lib().dvg_stem_gemm, _p(vec), vec.stride(0), _p(w_kn),
         kp, _p(scale), _p(shift), _p(out), out.stride(0), m, n, k, period, act, slope, _stream())
    return out


def transpose2d(a: torch.Tensor) -> torch.Tensor:
    """[R][C] -> contiguous [C][R] with the LDS-tiled layout kernel (a (1,R,C) 'NCHW' -> 'NHWC' pass)."""
    _dev_f32(a, "transpose2d")
    a = a if a.is_contiguous() else a.contiguous()
    r, c = a.shape
    out = torch.empty((c, r), device=a.device, dtype=torch.float32)
    check(lib().dvg_nchw_to_nhwc(_p(a), _p(out), 1, r, c, 1, _stream()), "transpose2d")
    return out


def colsum(a, out=None, accumulate=False):
    a = a if a.is_contiguous() else a.contiguous()
    rows, c = a.shape
    if out is None:
        out, accumulate = torch.empty(c, device=a.device, dtype=torch.float32), False
    check(lib().dvg_colsum(_p(a), _p(out), rows, c, int(accumulate), _stream()), "colsum")
    return out


GEMM_TN_MAX_ROWS = 512     # above: two transposes + the NT kernel (its K = rows spread over lanes and split-K suit long sums)


def gemm_tn(a, b, out=None, accumulate=False, colsums=(), colsum_accumulate=True):
    """out (M, N) (+)= a^T b for a (R, M), b (R, N) - a dense layer's weight gradient dY^T X over the rows of a BPTT pass - in one
    launch (dvg_gemm_tn); `colsums`: up to two (M,) buffers that receive (colsum_accumulate: are added) the column sums of `a`
    (the bias gradients).  Returns out."""
    _dev_f32(a, "gemm_tn.a")
    _dev_f32(b, "gemm_tn.b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise RuntimeError(f"gemm_tn: a {tuple(a.shape)} and b {tuple(b.shape)} must share their rows")
    a = a if a.stride(1) == 1 else a.contiguous()
    b = b if b.stride(1) == 1 else b.contiguous()
    r, m = a.shape
    n = b.shape[1]
    colsums = [c for c in colsums if c is not None]
    if len(colsums) > 2 or any(c.numel() != m or not c.is_contiguous() for c in colsums):
        raise RuntimeError("gemm_tn: at most two contiguous column-sum buffers of a.shape[1] entries")
    if out is None:
        out, accumulate = torch.empty((m, n), device=a.device, dtype=torch.float32), False
    elif tuple(out.shape) != (m, n) or out.stride(1) != 1:
        raise RuntimeError(f"gemm_tn: out must be ({m},{n}) with unit column stride")
    if r > GEMM_TN_MAX_ROWS:
        gemm_nt(transpose2d(a), transpose2d(b), None, None, out=out, accumulate=accumulate)
        for c in colsums:
            colsum(a, out=c, accumulate=colsum_accumulate)
        return out
    _run("gemm_tn", 2.0 * r * m * n, 4.0 * (r * m + r * n + m * n), lib().dvg_gemm_tn, _p(a), _p(b), _p(out),
         _p(colsums[0]) if colsums else None, _p(colsums[1]) if len(colsums) > 1 else None, r, m, n, a.stride(0), b.stride(0),
         out.stride(0), int(accumulate), int(colsum_accumulate), _stream())
    return out


def lstm_gates_bwd(dh, dc, gates, c_prev, c_new):
    b, hid = c_prev.shape
    dh = None if dh is None else (dh if dh.is_contiguous() else dh.contiguous())
    dc = None if dc is None else (dc if dc.is_contiguous() else dc.contiguous())
    dG = torch.empty((b, 4 * hid), device=c_prev.device, dtype=torch.float32)
    dcp = torch.empty_like(c_prev)
    check(lib().dvg_lstm_gates_bwd(_p(dh), _p(dc), _p(gates), _p(c_prev), _p(c_new), _p(dG), _p(dcp), b, hid,
                                   _stream()), "lstm_gates_bwd")
    return dG, dcp
