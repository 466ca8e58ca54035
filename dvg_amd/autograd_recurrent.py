"""torch.autograd glue of the dense / recurrent part of the training path: nn.Linear (+ activation), nn.LSTMCell, and the
teacher-forced LSTM sequence (lstm.py:42-72 driven by train.py:175-198,213-226) - forward and backward are kernels of
libdvg_hip.so (dense.hip).  Split from autograd.py in r06; that module re-exports these names and owns the shared machinery
(`_sink`, the deferred dense weight-gradient queues, the switches tests toggle), reached here through `_ag.` at call time."""
from __future__ import annotations

import torch

from . import autograd as _ag
from . import ops
from .ops import ACT_NONE

class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = _ag._c(x)
        y = ops.gemm_nt(x, weight.detach(), None, bias.detach() if bias is not None else None, act=act)
        ctx.save_for_backward(x, weight, y)
        ctx.params = (weight, bias)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dpre = ops.act_bwd(dy, y, ctx.act) if ctx.act != ACT_NONE else _ag._c(dy)
        dx = ops.gemm_nt(dpre, _ag._transposed(ctx.params[0]), None, None) if ctx.needs_input_grad[0] else None
        s_w = _ag._sink(ctx.params[0], ctx.needs_input_grad[1])
        s_b0 = _ag._sink(ctx.params[1], ctx.needs_input_grad[2]) if ctx.has_bias else None
        if _ag.DENSE_BATCH > 1 and s_w is not None and (s_b0 is not None or not ctx.has_bias):
            _ag._dense_wgrad((s_w,), (s_b0,), dpre, (x,))
            return dx, None, None, None
        s_b = _ag._sink(ctx.params[1], ctx.needs_input_grad[2]) if ctx.has_bias else None
        dW = ops.gemm_tn(dpre, x, out=s_w, accumulate=s_w is not None, colsums=(s_b,))
        db = ops.colsum(dpre) if ctx.has_bias and s_b is None else None
        return dx, (None if s_w is not None else dW), db, None


def linear_autograd(x, weight, bias, act):
    return _Linear.apply(x, weight, bias, act)


class _LSTMCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, c, w_ih, w_hh, b_ih, b_hh):
        x, h, c = _ag._c(x), _ag._c(h), _ag._c(c)
        h2, c2, gates = ops.lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh, want_gates=True)
        ctx.save_for_backward(x, h, c, w_ih, w_hh, gates, c2)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        return h2, c2

    @staticmethod
    def backward(ctx, dh2, dc2):
        x, h, c, w_ih, w_hh, gates, c2 = ctx.saved_tensors
        dG, dc = ops.lstm_gates_bwd(dh2, dc2, gates, c, c2)
        dx = ops.gemm_nt(dG, _ag._transposed(ctx.params[0]), None, None) if ctx.needs_input_grad[0] else None
        dh = ops.gemm_nt(dG, _ag._transposed(ctx.params[1]), None, None) if ctx.needs_input_grad[1] else None
        ng = ctx.needs_input_grad
        sinks = [_ag._sink(p, n) for p, n in zip(ctx.params, ng[3:7])]
        if _ag.DENSE_BATCH > 1 and all(t is not None for t in sinks):
            _ag._dense_wgrad((sinks[0], sinks[1]), (sinks[2], sinks[3]), dG, (x, h))
            return dx, dh, (dc if ng[2] else None), None, None, None, None
        dw_ih = ops.gemm_tn(dG, x, out=sinks[0], accumulate=sinks[0] is not None)
        dw_hh = ops.gemm_tn(dG, h, out=sinks[1], accumulate=sinks[1] is not None)
        dbs = []
        db = None
        for s_b in sinks[2:]:
            if s_b is not None:
                ops.colsum(dG, out=s_b, accumulate=True)
                dbs.append(None)
            else:
                db = ops.colsum(dG) if db is None else db
                dbs.append(db)
        return (dx, dh, (dc if ng[2] else None), None if sinks[0] is not None else dw_ih,
                None if sinks[1] is not None else dw_hh, dbs[0], dbs[1])


def lstm_cell_autograd(x, h, c, w_ih, w_hh, b_ih, b_hh):
    return _LSTMCell.apply(x, h, c, w_ih, w_hh, b_ih, b_hh)


_ZERO_STATES = {}


def _zero_state(b, h, dev):
    """One shared all-zero (b, h) tensor per shape: the initial (h, c) of every sequence, read only (a fill launch per closure
    otherwise).  Created eagerly; a first call during a hipGraph capture allocates from the graph's pool instead and is not
    cached."""
    key = (str(dev), b, h)
    z = _ZERO_STATES.get(key)
    if z is None:
        z = torch.zeros((b, h), device=dev)
        if not torch.cuda.is_current_stream_capturing():
            _ZERO_STATES[key] = z
    return z


class _LSTMSequence(torch.autograd.Function):
    """lstm.lstm (lstm.py:42-72) over a whole TEACHER-FORCED sequence (train.py:213-222,181-188: step i's input is the
    encoding of the ground-truth frame x[i-1], never a prediction), from the zero state of `init_hidden()`:
        e_t = W_e x_t + b_e;   per layer l: (h^l_t, c^l_t) = LSTMCell_l(h^{l-1}_t, (h^l_{t-1}, c^l_{t-1}));   y_t = tanh(W_o h^L_t + b_o)
    Only W_hh h_{t-1} depends on the recurrence.  Everything else runs as ONE GEMM over the S x B rows of the sequence, layer
    by layer (layer l's recurrence only needs its own past and layer l-1's complete output sequence): the embedding, every
    cell's input half W_ih (.) + b_ih + b_hh, the output head - forward - and their data and weight gradients - backward;
    per time step and layer one dvg_lstm_cell_pre launch forward and one dvg_lstm_cell_bwd launch backward (gate gradients
    + the recurrent hand-over dG W_hh).  2 L S + 4 launches forward instead of (L + 2) S; per step results identical to the
    step-by-step path up to fp32 summation order (the input half of the gates is added as one term).
    x (S*B, in).  params = (W_e, b_e, [W_ih, W_hh, b_ih, b_hh] x L, W_o, b_o).  Returns y (S*B, out)."""

    @staticmethod
    def forward(ctx, x, S, *params):
        L = (len(params) - 4) // 4
        we, be, wo, bo = params[0], params[1], params[-2], params[-1]
        x = _ag._c(x)
        rows, B = x.shape[0], x.shape[0] // S
        H = we.shape[0]
        dev = x.device
        e = ops.gemm_nt(x, we.detach(), None, be.detach())
        zero = _zero_state(B, H, dev)
        inp, saved = e, []
        for l in range(L):
            wih, whh, bih, bhh = params[2 + 4 * l: 6 + 4 * l]
            pre = ops.gemm_nt(inp, wih.detach(), None, bih.detach() + bhh.detach())        # (S*B, 4H): input half + both biases
            hs, cs = torch.empty((rows, H), device=dev), torch.empty((rows, H), device=dev)
            gs = torch.empty((rows, 4 * H), device=dev)
            hp, cp = zero, zero
            for t in range(S):
                sl = slice(t * B, (t + 1) * B)
                ops.lstm_cell_pre(pre[sl], hp, cp, whh.detach(), hs[sl], cs[sl], gs[sl])
                hp, cp = hs[sl], cs[sl]
            saved += [inp, hs, cs, gs]
            inp = hs
        y = ops.gemm_nt(inp, wo.detach(), None, bo.detach(), act=ops.ACT_TANH)
        ctx.save_for_backward(x, y, zero, *saved)
        ctx.params, ctx.meta = params, (S, B, H, L)
        ctx.param_versions = tuple(p._version for p in params)   # backward re-reads the weights: see the check there
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, zero, *saved = ctx.saved_tensors
        params = ctx.params
        # The parameters are kept as objects (their .grad buffers are the in-place sinks), not through save_for_backward, so
        # autograd's own version check does not cover them: an optimiser step between this node's forward and backward
        # would make the transposed weights below the NEW ones.
        if tuple(p._version for p in params) != ctx.param_versions:
            raise RuntimeError("_LSTMSequence.backward: an LSTM parameter was modified in place after the forward pass")
        S, B, H, L = ctx.meta
        ng = ctx.needs_input_grad            # (x, S, *params)
        pg = list(ng[2:])
        grads = [None] * len(params)

        def acc_wb(i, d, inp, bias=()):      # dW_i = d^T inp and db_j = column sums of d (j in bias): ONE launch when the
            sinks = []                       # gradients go into the parameters' .grad buffers (in-place sinks)
            plain = None
            for j in bias:
                if not pg[j]:
                    continue
                s_ = _ag._sink(params[j], True)
                if s_ is not None:
                    sinks.append(s_)
                else:
                    plain = ops.colsum(d) if plain is None else plain
                    grads[j] = plain
            if pg[i]:
                s_ = _ag._sink(params[i], True)
                g = ops.gemm_tn(d, inp, out=s_, accumulate=s_ is not None, colsums=sinks)
                grads[i] = None if s_ is not None else g
            else:
                for s_ in sinks:
                    ops.colsum(d, out=s_, accumulate=True)
        dpre = ops.act_bwd(_ag._c(dy), y, ops.ACT_TANH)
        top = saved[4 * (L - 1) + 1]
        acc_wb(len(params) - 2, dpre, top, [len(params) - 1])
        dh_all = ops.gemm_nt(dpre, _ag._transposed(params[-2]), None, None)          # d h^L_t for every t, (S*B, H)
        dev = x.device
        for l in reversed(range(L)):
            inp, hs, cs, gs = saved[4 * l: 4 * l + 4]
            wih, whh = params[2 + 4 * l], params[3 + 4 * l]
            whh_t = _ag._transposed(whh)                                              # [H][4H]
            dG = torch.empty((S * B, 4 * H), device=dev)
            dcb = [torch.empty((B, H), device=dev), torch.empty((B, H), device=dev)]
            dhb = [torch.empty((B, H), device=dev), torch.empty((B, H), device=dev)]
            dh_rec = dc = None
            for t in reversed(range(S)):
                sl = slice(t * B, (t + 1) * B)
                c_prev = cs[(t - 1) * B: t * B] if t > 0 else zero
                dcp, dhp = dcb[t & 1], (dhb[t & 1] if t > 0 else None)
                ops.lstm_cell_bwd(dh_all[sl], dh_rec, dc, gs[sl], c_prev, cs[sl], whh_t, dG[sl], dcp, dhp)
                dh_rec, dc = dhp, dcp
            acc_wb(2 + 4 * l, dG, inp, [4 + 4 * l, 5 + 4 * l])
            if S > 1:                           # h_{-1} = 0: the first step contributes nothing to dW_hh
                acc_wb(3 + 4 * l, dG[B:], hs[:(S - 1) * B])
            if l > 0 or pg[0] or pg[1] or ng[0]:
                dh_all = ops.gemm_nt(dG, _ag._transposed(wih), None, None)          # gradient w.r.t. this layer's input sequence
        de = dh_all
        acc_wb(0, de, x, [1])
        dx = ops.gemm_nt(de, _ag._transposed(params[0]), None, None) if ng[0] else None
        return (dx, None) + tuple(grads)


def lstm_sequence_autograd(x, S, params):
    return _LSTMSequence.apply(x, S, *params)
