"""Deterministic, platform-independent parameter / input generators for the parity tests.
*** TEST INFRASTRUCTURE ONLY *** (see oracle/dvg_oracle.py).

Everything is drawn from numpy's PCG64 (`numpy.random.default_rng(seed)`), never from torch
RNG streams, so the container that imports the reference (golden generation) and the GPU box
(parity tests) regenerate bit-identical weights from a seed instead of shipping them.

Weights are scaled ~ 1.2/sqrt(fan_in) rather than the reference's N(0, 0.02) init
(utils.py:304-311): with eval-mode BatchNorm and arbitrary running statistics the N(0,0.02)
init makes a 14-layer VGG collapse towards a constant, which would make parity checks
vacuous.  `init_weights_like_reference` reproduces the reference distribution where the
distribution itself matters (bench.py, train.py).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch


def _fan_in(shape, transposed: bool) -> int:
    if len(shape) == 4:
        # Conv2d (out,in,kh,kw); ConvTranspose2d (in,out,kh,kw): every output sums over in*kh*kw/stride^2
        cin = shape[0] if transposed else shape[1]
        return int(cin * shape[2] * shape[3])
    return int(shape[-1])


def fill_state_dict(template: "OrderedDict[str, torch.Tensor]", seed: int, transposed_keys=()) -> "OrderedDict":
    """Fill a state_dict-shaped template (key order matters!) with seeded values."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    keys = list(template.keys())
    bn_prefixes = {k[: -len(".running_mean")] for k in keys if k.endswith(".running_mean")}
    for k in keys:
        shape = tuple(template[k].shape)
        prefix = k.rsplit(".", 1)[0]
        if k.endswith("num_batches_tracked") or k.endswith("variational_params_initialized"):
            out[k] = torch.zeros(shape, dtype=torch.long)
            continue
        if k.endswith(".running_mean"):
            v = rng.normal(0.0, 0.1, shape)
        elif k.endswith(".running_var"):
            v = rng.uniform(0.7, 1.3, shape)
        elif prefix in bn_prefixes and k.endswith(".weight"):
            v = rng.normal(1.0, 0.1, shape)
        elif prefix in bn_prefixes and k.endswith(".bias"):
            v = rng.normal(0.0, 0.1, shape)
        elif len(shape) >= 2:
            tr = k in transposed_keys
            fi = _fan_in(shape, tr)
            if tr and len(shape) == 4 and shape[2] == 4 and "upc1" not in k:
                fi //= 4  # stride-2 transposed conv: each output sees a 2x2 subset of the 4x4 taps
            v = rng.normal(0.0, 1.2 / np.sqrt(fi), shape)
        else:
            v = rng.normal(0.0, 0.05, shape)
        out[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))
    return out


def decoder_transposed_keys(template, family: str) -> tuple:
    """Exact keys of the ConvTranspose2d weights of a decoder template: for the vgg family only the
    stem (upc1.0) and the final 3x3 layer, for dcgan every 4-D weight."""
    four_d = [k for k, v in template.items() if v.dim() == 4]
    if family == "dcgan":
        return tuple(four_d)
    last = max(int(k[3]) for k in template if k.startswith("upc"))
    return tuple(["upc1.0.weight"] + [k for k in four_d if k.startswith(f"upc{last}.") and template[k].shape[1] <= 4])


def frames(seed: int, batch: int, nc: int, res: int, lo=0.0, hi=1.0) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(lo, hi, (batch, nc, res, res)).astype(np.float32))


def normal(seed: int, *shape, scale=1.0) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.normal(0.0, scale, shape)).astype(np.float32))


def gp_state(seed: int, D: int = 90, M: int = 40, trained: bool = True):
    """A GP state_dict (gpytorch 0.3.x key names) and a likelihood state_dict.  `trained` perturbs
    the variational parameters away from the prior so that S' != K_ZZ^-1 (the regime where the
    whitened and un-whitened parameterisations differ)."""
    rng = np.random.default_rng(seed)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))  # noqa: E731
    sd = OrderedDict()
    sd["variational_strategy.inducing_points"] = f32(rng.uniform(-1.0, 1.0, (D, M, 1)))
    sd["variational_strategy.variational_params_initialized"] = torch.tensor(1)
    sd["variational_strategy.variational_distribution.variational_mean"] = f32(rng.normal(0, 0.3, (D, M)))
    chol = np.tril(rng.normal(0, 0.05, (D, M, M)))
    chol[:, np.arange(M), np.arange(M)] = rng.uniform(0.3, 1.0, (D, M))
    sd["variational_strategy.variational_distribution.chol_variational_covar"] = f32(chol)
    sd["mean_module.constant"] = f32(rng.normal(0, 0.1, (D, 1)))
    sd["covar_module.raw_outputscale"] = f32(rng.normal(0, 0.3, (D,)))
    sd["covar_module.base_kernel.raw_lengthscale"] = f32(rng.normal(-0.5, 0.3, (D, 1, 1)))
    lik = OrderedDict()
    lik["noise_covar.raw_noise"] = f32(rng.normal(-2.0, 0.3, (D, 1)))
    if not trained:
        sd["variational_strategy.variational_params_initialized"] = torch.tensor(0)
    return sd, lik
