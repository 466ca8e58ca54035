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
        h_in = linear(self.embed, input.reshape(-1, self.input_size))
        for i in range(self.n_layers):
            h, c = self.hidden[i]
            if h.device != dev:  # state created before .cuda(): the reference re-creates it per sequence
                h, c = h.to(dev), c.to(dev)
            self.hidden[i] = cell(self.lstm[i], h_in, (h, c))
            h_in = self.hidden[i][0]
        return h_in


class lstm(_Recurrent):
    """lstm.lstm (lstm.py:42-72): Linear -> n_layers x LSTMCell -> Linear + Tanh."""

    def __init__(self, input_size, output_size, hidden_size, n_layers, batch_size):
        super().__init__(input_size, output_size, hidden_size, n_layers, batch_size)
        self.output = nn.Sequential(nn.Linear(hidden_size, output_size), nn.Tanh())
        self.hidden = self.init_hidden()

    def forward(self, input):
        return linear(self.output[0], self._trunk(input), ACT_TANH)


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
