"""models.lstm counterpart: the per-timestep recurrent latent predictor.

`lstm` (reference lstm.py:42-72) and `gaussian_lstm` (lstm.py:140-175) keep the
reference's constructor signature, attribute tree (embed / lstm.{i} / output.0 |
mu_net / logvar_net: state_dict-compatible) and the externally assigned, module-held
`hidden` state (train.py:150,178,206,263).  The math runs in libdvg_hip.so:
one small-M GEMM per nn.Linear and one `dvg_lstm_cell` launch per nn.LSTMCell.

Deviation from the reference: `init_hidden()` allocates on the module's own device
instead of hard-calling `.cuda()` (lstm.py:61-62), so the class can be constructed on a
CPU-only host; the forward itself requires a GPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_NONE, ACT_TANH


def _grad_on(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def linear(mod: nn.Linear, x: torch.Tensor, act: int = ACT_NONE) -> torch.Tensor:
    """act(x W^T + b) through dvg_gemm_nt_bias_act (nn.Linear of lstm.py:50,53-55)."""
    if _grad_on(x, mod.weight, mod.bias):
        from ..autograd import linear_autograd
        return linear_autograd(x, mod.weight, mod.bias, act)
    return ops.gemm_nt(x, mod.weight.detach(), None, mod.bias.detach() if mod.bias is not None else None, act=act)


def cell(mod: nn.LSTMCell, x: torch.Tensor, hc):
    h, c = hc
    if _grad_on(x, h, c, mod.weight_ih):
        from ..autograd import lstm_cell_autograd
        return lstm_cell_autograd(x, h, c, mod.weight_ih, mod.weight_hh, mod.bias_ih, mod.bias_hh)
    return ops.lstm_cell(x, h, c, mod.weight_ih, mod.weight_hh, mod.bias_ih, mod.bias_hh)


def reparameterize(mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
    """z = eps * exp(0.5*logvar) + mu, eps ~ N(0,1) from the global torch RNG (lstm.py:161-164)."""
    eps = torch.randn_like(logvar)
    return eps * torch.exp(0.5 * logvar) + mu


_FOLD_CACHE = {}   # id(module) -> (weakref, versions, w_x, bias)


def folded_first_cell(embed: nn.Linear, cell0: nn.LSTMCell):
    """W_x = W_ih W_e (4H, Kxp: rows zero-padded to a multiple of 4 floats) and bias = W_ih b_e + b_ih + b_hh of the first
    cell with the embedding folded in (see dvg_lstm_cell_x), cached per parameter version; None when the shapes do not fit
    the kernel.  Both products run through dvg_gemm_nt_bias_act."""
    import weakref
    kx, hid = embed.in_features, cell0.hidden_size
    if kx % 2 or kx > 128 or hid % 64 or embed.out_features != cell0.input_size or embed.bias is None:
        return None
    ps = (embed.weight, embed.bias, cell0.weight_ih, cell0.bias_ih, cell0.bias_hh)
    key = tuple((p.data_ptr(), p._version) for p in ps)
    hit = _FOLD_CACHE.get(id(cell0))
    if hit is not None and hit[0]() is cell0 and hit[1] == key:
        return hit[2], hit[3]
    with torch.no_grad():
        w_ih = cell0.weight_ih.detach()
        kxp = (kx + 3) // 4 * 4
        w_x = torch.zeros((4 * hid, kxp), device=w_ih.device, dtype=torch.float32)
        ops.gemm_nt(w_ih, embed.weight.detach().t().contiguous(), None, None, out=w_x[:, :kx])
        bias = ops.gemm_nt(w_ih, embed.bias.detach().view(1, -1), None, None).view(-1)
        bias = (bias + cell0.bias_ih.detach() + cell0.bias_hh.detach()).contiguous()
    if len(_FOLD_CACHE) > 64:
        _FOLD_CACHE.clear()
    _FOLD_CACHE[id(cell0)] = (weakref.ref(cell0), key, w_x, bias)
    return w_x, bias


class _Recurrent(nn.Module):
    def __init__(self, input_size, output_size, hidden_size, n_layers, batch_size):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.hidden_size = hidden_size
        self.batch_size = batch_size
        self.n_layers = n_layers
        self.embed = nn.Linear(input_size, hidden_size)
        self.lstm = nn.ModuleList([nn.LSTMCell(hidden_size, hidden_size) for _ in range(n_layers)])

    def init_hidden(self):
        dev = self.embed.weight.device
        return [(torch.zeros(self.batch_size, self.hidden_size, device=dev),
                 torch.zeros(self.batch_size, self.hidden_size, device=dev)) for _ in range(self.n_layers)]

    def _trunk(self, input):
        dev = self.embed.weight.device
        x = input.reshape(-1, self.input_size)
        first = 0
        h_in = None
        if not _grad_on(x, self.embed.weight, self.lstm[0].weight_ih, self.hidden[0][0], self.hidden[0][1]):
            fold = folded_first_cell(self.embed, self.lstm[0])
            if fold is not None and x.is_cuda and x.data_ptr() % 8 == 0 and x.stride(0) % 2 == 0 and x.stride(1) == 1:
                # inference: embed folded into the first cell (one launch and 3/4 of its x-side FLOPs less per step)
                h, c = self.hidden[0]
                if h.device != dev:
                    h, c = h.to(dev), c.to(dev)
                self.hidden[0] = ops.lstm_cell_x(x, h, c, fold[0], self.lstm[0].weight_hh, fold[1])
                h_in, first = self.hidden[0][0], 1
        if h_in is None:
            h_in = linear(self.embed, x)
        for i in range(first, self.n_layers):
            h, c = self.hidden[i]
            if h.device != dev:  # state created before .cuda(): the reference re-creates it per sequence
                h, c = h.to(dev), c.to(dev)
            self.hidden[i] = cell(self.lstm[i], h_in, (h, c))
            h_in = self.hidden[i][0]
        return h_in

    def step_state_only(self, input) -> None:
        """Advance the hidden state on `input` WITHOUT computing the output: what the rollouts do on conditioning frames,
        where the reference calls `frame_predictor(h)` and discards the result (generate_frames.py:125,162; train.py:276)."""
        self._trunk(input)


class lstm(_Recurrent):
    """lstm.lstm (lstm.py:42-72): Linear -> n_layers x LSTMCell -> Linear + Tanh."""

    def __init__(self, input_size, output_size, hidden_size, n_layers, batch_size):
        super().__init__(input_size, output_size, hidden_size, n_layers, batch_size)
        self.output = nn.Sequential(nn.Linear(hidden_size, output_size), nn.Tanh())
        self.hidden = self.init_hidden()

    def forward(self, input):
        return linear(self.output[0], self._trunk(input), ACT_TANH)


def sequence_applies(mod) -> bool:
    """forward_sequence's kernels: rnn_size 256 (train.py:37; dvg_lstm_cell_bwd maps one gate to one 256-wide K slice)."""
    return isinstance(mod, lstm) and mod.hidden_size == 256 and mod.lstm[0].input_size == 256


def forward_sequence(mod: "lstm", hseq: torch.Tensor) -> torch.Tensor:
    """`[mod(h) for h in hseq]` for a TEACHER-FORCED sequence hseq (S, B, in) - the inputs of all steps exist up front
    (train.py:181-188,213-222) - starting from the zero state of `init_hidden()`: (S, B, out).  One GEMM per non-recurrent
    product over all S x B rows (autograd._LSTMSequence).
    REQUIREMENT ON CALLERS: `mod.hidden` is NOT advanced (the step-by-step module leaves the state after step S there) - a
    caller must not read `mod.hidden` after this call; it is set to None so that such a read fails instead of returning the
    stale pre-sequence state.  The closures that use this re-create it per sequence (train.py:178,206) and never read it."""
    from ..autograd import lstm_sequence_autograd
    S, B = hseq.shape[0], hseq.shape[1]
    params = [mod.embed.weight, mod.embed.bias]
    for cell_ in mod.lstm:
        params += [cell_.weight_ih, cell_.weight_hh, cell_.bias_ih, cell_.bias_hh]
    params += [mod.output[0].weight, mod.output[0].bias]
    y = lstm_sequence_autograd(hseq.reshape(S * B, -1), S, params)
    mod.hidden = None
    return y.view(S, B, -1)


class gaussian_lstm(_Recurrent):
    """lstm.gaussian_lstm (lstm.py:140-175): same trunk, mu / logvar heads, reparameterised z."""

    def __init__(self, input_size, output_size, hidden_size, n_layers, batch_size):
        super().__init__(input_size, output_size, hidden_size, n_layers, batch_size)
        self.mu_net = nn.Linear(hidden_size, output_size)
        self.logvar_net = nn.Linear(hidden_size, output_size)
        self.hidden = self.init_hidden()

    def reparameterize(self, mu, logvar):
        return reparameterize(mu, logvar)

    def forward(self, input):
        h_in = self._trunk(input)
        mu = linear(self.mu_net, h_in)
        logvar = linear(self.logvar_net, h_in)
        return self.reparameterize(mu, logvar), mu, logvar
