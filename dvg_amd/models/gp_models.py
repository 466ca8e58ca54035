"""models.gp_models counterpart: the batched sparse variational GP "trigger".

`GPRegressionLayer1(num_dims=90, num_inducing_points=40)` keeps the reference
constructor (gp_models.py:10-11) and the gpytorch-0.3.x state_dict key names
(SURVEY.md §8(b)), but contains no gpytorch: the arithmetic that gpytorch's
WhitenedVariationalStrategy / GaussianLikelihood / MultivariateNormal /
VariationalELBO performed for the reference call sites (train.py:101-112,225-232,
283-284; generate_frames.py:67-72,131,170,229,273,291) is stated in DESIGN.md
("GP: equations of record") and runs as ONE `dvg_gp_predict` launch per call:
RBF assembly, both Cholesky factorisations, solves, predictive moments, KL and the
reparameterised sample, one workgroup per latent dimension.

`GaussianLikelihood` and `VariationalELBO` are the minimal stand-ins for
`gpytorch.likelihoods.GaussianLikelihood(batch_size=D)` (train.py:102) and
`gpytorch.mlls.VariationalELBO(likelihood, gp, num_data, combine_terms=True)`
(train.py:112).  Parity for this module is *unpinned* (no gpytorch, no reference
tests): it is validated against oracle/dvg_oracle.py's fp64 restatement.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

JITTER = 1e-3        # lazy_tensor.add_jitter() default of gpytorch 0.3.x
NOISE_FLOOR = 1e-4   # GaussianLikelihood noise constraint GreaterThan(1e-4)
# First-call initialisation of the variational distribution (an UNTRAINED GP's start only; a loaded state with
# variational_params_initialized = 1 never runs it): L_S <- chol((K_ZZ + INIT_JITTER_TERMS * JITTER * I)^-1).
# [recalled, unverifiable here - gpytorch is not installable]  gpytorch 0.3.x's WhitenedVariationalStrategy builds its prior as
# MultivariateNormal(mean, K_ZZ.add_jitter()) and `initialize_variational_dist` inverts `prior.lazy_covariance_matrix
# .add_jitter()`: in the releases where BOTH calls add the default 1e-3 (0.3.2 - 0.3.6 as recalled) the inverse is of K_ZZ +
# 2e-3 I, in 0.3.0 / 0.3.1 (initialisation from `prior_dist.covariance_matrix` directly) of K_ZZ + 1e-3 I.  1 = the GP starts
# EXACTLY at its prior under the predictive equations of record (KL = 0, variance = outputscale: what tests/test_oracle_gp.py
# pins); 2 = the recalled later behaviour (start KL = 1/2 sum_i [r_i - 1 - ln r_i], r_i = (lam_i + j) / (lam_i + 2 j) over K_ZZ's eigenvalues:
# ~0.5 nats per latent dim, predictive variance within a few per cent of the prior's).  One
# constant, both tested (tests/test_oracle_gp.py, tests/test_gpu_parity.py); the oracle mirrors it (oracle.gp_prior_init).
INIT_JITTER_TERMS = 1

# gpytorch >= 0.3.3 registers its parameter constraints as modules whose bounds are buffers in the state_dict; the arithmetic
# here hard-wires exactly these (softplus onto (0, inf) for outputscale / lengthscale, softplus + 1e-4 for the noise)
_EXPECTED_BOUNDS = {
    "covar_module.raw_outputscale_constraint": (0.0, math.inf),
    "covar_module.base_kernel.raw_lengthscale_constraint": (0.0, math.inf),
    "noise_covar.raw_noise_constraint": (NOISE_FLOOR, math.inf),
}


def _tolerant_state_dict(module: nn.Module, state_dict, what: str):
    """A real gpytorch-0.3.x state_dict -> (state_dict this module loads strictly, report).  Accepted and REPORTED, never
    silently dropped (generate_frames.py:67-72 of the reference loads `gp_layer` / `likelihood` state_dicts):
      * shape variants of a key with the right number of elements ((90,1,1) / (90,1) / (90,) for the per-dim scalars, a
        (90,1,40) variational mean, ...) are reshaped;
      * `<param>_constraint.lower_bound / upper_bound` buffers are checked against the bounds this implementation hard-wires
        and left out; DIFFERENT bounds raise (the arithmetic would differ);
      * a missing `variational_params_initialized` is read as 1 when variational parameters are present (a trained state).
    Anything else unknown stays in the dict, so that load_state_dict(strict=True) still names it."""
    own = module.state_dict()
    out, report = type(state_dict)(), {"reshaped": [], "constraints": [], "assumed": []}
    for k, v in state_dict.items():
        base, _, leaf = k.rpartition(".")
        if leaf in ("lower_bound", "upper_bound") and base in _EXPECTED_BOUNDS:
            want = _EXPECTED_BOUNDS[base][0 if leaf == "lower_bound" else 1]
            got = float(torch.as_tensor(v).reshape(-1)[0])
            if not (got == want or abs(got - want) <= 1e-6 * abs(want)):       # (a float32 buffer holds 1e-4 as 9.9999997e-05)
                raise RuntimeError(f"{what}: state_dict carries {k} = {got}, this implementation hard-wires {want} "
                                   "(softplus-positive kernel hyper-parameters, noise >= 1e-4): results would differ")
            report["constraints"].append(f"{k} = {got:g}")
            continue
        if k in own and torch.is_tensor(v) and tuple(v.shape) != tuple(own[k].shape) and v.numel() == own[k].numel():
            report["reshaped"].append(f"{k}: {tuple(v.shape)} -> {tuple(own[k].shape)}")
            v = v.reshape(own[k].shape)
        out[k] = v
    flag = "variational_strategy.variational_params_initialized"
    if flag in own and flag not in out and "variational_strategy.variational_distribution.variational_mean" in out:
        out[flag] = torch.tensor(1)
        report["assumed"].append(f"{flag} = 1 (absent; variational parameters present)")
    return out, report


def _announce(what: str, report: dict) -> None:
    if any(report.values()):
        import sys
        print(f"{what}.load_state_dict: " + "; ".join(f"{k}: {', '.join(v)}" for k, v in report.items() if v), file=sys.stderr)


class _Holder(nn.Module):
    """Parameter container (never called)."""


class GPPrediction:
    """What the reference code reads off a gpytorch MultivariateNormal: `.mean` (D,B),
    `.variance` (D,B), `.rsample()` (D,B), `.covariance_matrix` (D,B,B).  Evaluation is lazy
    so that `likelihood(gp_layer(x)).rsample()` is a single kernel launch."""

    def __init__(self, layer: "GPRegressionLayer1", h: torch.Tensor, noise=None, training=False, raw_noise=None):
        self._layer, self._h, self._noise, self._training = layer, h, noise, training
        self._raw_noise = raw_noise        # inference: the likelihood's RAW noise parameter (soft-plus'ed in the kernel)
        self._res = None
        self._cov = None

    # -- evaluation ------------------------------------------------------------------
    def _run(self, eps=None, want_cov=False):
        lay = self._layer
        if self._training and lay._grad_needed(self._h):
            from ..autograd import gp_train_autograd
            return gp_train_autograd(lay, self._h, self._noise)
        vs = lay.variational_strategy
        if self._noise is None and not torch.is_grad_enabled():
            # inference: raw hyper-parameters straight into the kernel (it soft-pluses them; no 90-element torch launches)
            return ops.gp_predict(self._h, vs.inducing_points, vs.variational_distribution.variational_mean,
                                  vs.variational_distribution.chol_variational_covar, lay.mean_module.constant,
                                  lay.covar_module.raw_outputscale, lay.covar_module.base_kernel.raw_lengthscale,
                                  noise=self._raw_noise, eps=eps, want_cov=want_cov, want_kl=self._training,
                                  train_mode=self._training, jitter=JITTER, raw_hypers=True)
        s, ell, c = lay.hypers()
        noise = self._noise
        if noise is None and self._raw_noise is not None:
            noise = F.softplus(self._raw_noise).reshape(-1) + NOISE_FLOOR
        return ops.gp_predict(self._h, vs.inducing_points, vs.variational_distribution.variational_mean,
                              vs.variational_distribution.chol_variational_covar, c, s, ell,
                              noise=noise, eps=eps, want_cov=want_cov, want_kl=self._training,
                              train_mode=self._training, jitter=JITTER)

    def _moments(self):
        if self._res is None:
            self._res = self._run()
        return self._res

    @property
    def mean(self):
        return self._moments()["mean"]

    @property
    def variance(self):
        return self._moments()["var"]

    @property
    def kl(self):
        return self._moments()["kl"]

    @property
    def covariance_matrix(self):
        if self._training:
            raise RuntimeError("train-mode prediction carries a diagonal covariance only (gpytorch 0.3.x)")
        if self._cov is None:
            r = self._run(want_cov=True)
            self._res, self._cov = r, r["cov"]
        return self._cov

    def rsample(self, base_samples: torch.Tensor = None):
        """mean + chol(Sigma) eps;  eps ~ N(0,I) (D,B) from the global torch RNG unless given."""
        d, b = self._layer.num_dims, self._h.shape[0]
        eps = base_samples if base_samples is not None else torch.randn(d, b, device=self._h.device)
        if self._training:
            m = self._moments()
            return m["mean"] + torch.sqrt(m["var"]) * eps
        r = self._run(eps=eps)
        if self._res is None:
            self._res = r
        return r["sample"]

    sample = rsample

    def with_noise(self, noise):
        return GPPrediction(self._layer, self._h, noise, self._training)

    def with_raw_noise(self, raw_noise):
        return GPPrediction(self._layer, self._h, None, self._training, raw_noise=raw_noise)


class GPRegressionLayer1(nn.Module):
    """D independent 1-D sparse variational GPs, M learnable inducing points each
    (gp_models.py:10-24): constant mean, ScaleKernel(RBFKernel), Cholesky variational
    distribution, whitened variational strategy."""

    def __init__(self, num_dims=90, num_inducing_points=40):
        super().__init__()
        D, M = num_dims, num_inducing_points
        self.num_dims, self.num_inducing_points = D, M
        vs = _Holder()
        vs.inducing_points = nn.Parameter(torch.rand(D, M, 1))                    # gp_models.py:13
        vd = _Holder()
        vd.variational_mean = nn.Parameter(torch.zeros(D, M))
        vd.chol_variational_covar = nn.Parameter(torch.eye(M).repeat(D, 1, 1))
        vs.variational_distribution = vd
        vs.register_buffer("variational_params_initialized", torch.tensor(0))
        self.variational_strategy = vs
        self.mean_module = _Holder()
        self.mean_module.constant = nn.Parameter(torch.zeros(D, 1))                # ConstantMean(batch_size=D)
        self.covar_module = _Holder()
        self.covar_module.raw_outputscale = nn.Parameter(torch.zeros(D))           # ScaleKernel(batch_size=D)
        self.covar_module.base_kernel = _Holder()
        self.covar_module.base_kernel.raw_lengthscale = nn.Parameter(torch.zeros(D, 1, 1))  # RBFKernel(batch_size=D)

    def load_state_dict(self, state_dict, *a, **k):
        """Strict on the gpytorch-0.3.x key set, tolerant of its shape variants and constraint buffers (_tolerant_state_dict);
        what was adapted is kept in `self.load_report` and printed once to stderr."""
        self._init_checked = False
        sd, self.load_report = _tolerant_state_dict(self, state_dict, "GPRegressionLayer1")
        _announce("GPRegressionLayer1", self.load_report)
        return super().load_state_dict(sd, *a, **k)

    # -- hyper-parameters ---------------------------------------------------------------
    def hypers(self):
        s = F.softplus(self.covar_module.raw_outputscale).reshape(-1)
        ell = F.softplus(self.covar_module.base_kernel.raw_lengthscale).reshape(-1)
        return s, ell, self.mean_module.constant.reshape(-1)

    def _grad_needed(self, h):
        return torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters()))

    @torch.no_grad()
    def initialize_variational_dist(self):
        """First-call initialisation of gpytorch's WhitenedVariationalStrategy: variational
        mean <- prior mean, chol_variational_covar <- chol((K_ZZ + jitter I)^-1) in fp64, so
        the GP starts at its prior."""
        vs = self.variational_strategy
        s, ell, c = [t.double() for t in self.hypers()]
        z = vs.inducing_points.squeeze(-1).double()
        diff = z.unsqueeze(-1) - z.unsqueeze(-2)
        kzz = s.view(-1, 1, 1) * torch.exp(-0.5 * diff * diff / ell.view(-1, 1, 1) ** 2)
        kzz = kzz + INIT_JITTER_TERMS * JITTER * torch.eye(z.shape[1], dtype=torch.float64, device=z.device)
        # one-off 40x40 fp64 inverse per latent dim at initialisation time (host of the path, not
        # on it): done on the CPU so that no BLAS/solver library is pulled onto the device
        ls = torch.linalg.cholesky(torch.linalg.inv(kzz.cpu())).to(z.device)
        vs.variational_distribution.chol_variational_covar.copy_(ls.to(torch.float32))
        vs.variational_distribution.variational_mean.copy_(c.view(-1, 1).expand_as(
            vs.variational_distribution.variational_mean).to(torch.float32))
        vs.variational_params_initialized.fill_(1)

    def ensure_initialized(self):
        if not getattr(self, "_init_checked", False):   # host-side flag: no device sync per call (graph-capturable)
            if not int(self.variational_strategy.variational_params_initialized.item()):
                self.initialize_variational_dist()
            self._init_checked = True

    def forward(self, x):
        """x: (D,B,1) as produced by `h.transpose(0,1).view(D,B,1)` (train.py:225) — a strided view
        that shares storage with h — or directly h (B,D)."""
        self.ensure_initialized()
        if x.dim() == 3:
            if x.shape[0] != self.num_dims or x.shape[2] != 1:
                raise RuntimeError(f"GP input must be ({self.num_dims},B,1), got {tuple(x.shape)}")
            h = x.squeeze(-1).transpose(0, 1)
        else:
            h = x
        if h.shape[1] != self.num_dims:
            raise RuntimeError(f"GP input must carry {self.num_dims} latent dims, got {tuple(h.shape)}")
        return GPPrediction(self, h, None, self.training)


class GaussianLikelihood(nn.Module):
    """gpytorch.likelihoods.GaussianLikelihood(batch_size=D): homoskedastic noise per latent
    dim, sigma^2 = softplus(raw_noise) + 1e-4; `likelihood(pred)` adds it to the predictive
    (co)variance (generate_frames.py:131,170)."""

    def __init__(self, batch_size=1):
        super().__init__()
        self.noise_covar = _Holder()
        self.noise_covar.raw_noise = nn.Parameter(torch.zeros(batch_size, 1))

    def load_state_dict(self, state_dict, *a, **k):
        sd, self.load_report = _tolerant_state_dict(self, state_dict, "GaussianLikelihood")
        _announce("GaussianLikelihood", self.load_report)
        return super().load_state_dict(sd, *a, **k)

    @property
    def noise(self):
        return F.softplus(self.noise_covar.raw_noise).reshape(-1) + NOISE_FLOOR

    def forward(self, pred: GPPrediction) -> GPPrediction:
        if not torch.is_grad_enabled():
            return pred.with_raw_noise(self.noise_covar.raw_noise)    # soft-plus + floor happen inside dvg_gp_predict
        return pred.with_noise(self.noise)

    def expected_log_prob(self, target, pred: GPPrediction):
        """E_q(f)[log N(y | f, sigma^2)] summed over the B points -> (D,)."""
        mean, var = pred.mean, pred.variance
        nz = self.noise.view(-1, 1)
        res = -0.5 * ((target - mean) ** 2 + var) / nz - 0.5 * torch.log(nz) - 0.5 * math.log(2 * math.pi)
        return res.sum(-1)


class VariationalELBO(nn.Module):
    """gpytorch.mlls.VariationalELBO(likelihood, model, num_data, combine_terms=True)
    (train.py:112): `mll(pred, target)` -> (D,) = E log-lik / B - KL / num_data."""

    def __init__(self, likelihood, model, num_data, combine_terms=True):
        super().__init__()
        self.likelihood, self.model, self.num_data, self.combine_terms = likelihood, model, num_data, combine_terms

    def forward(self, pred: GPPrediction, target: torch.Tensor):
        b = target.shape[-1]
        if (self.combine_terms and pred._training and pred._noise is None and pred._raw_noise is None and
                target.is_cuda and target.dim() == 2 and isinstance(self.likelihood, GaussianLikelihood)):
            # the reference's call sites (train.py:164-169,225-226): one launch forward, one backward (dvg_gp_elbo)
            from ..gp_autograd import gp_elbo
            return gp_elbo(pred.mean, pred.variance, pred.kl, target, self.likelihood.noise_covar.raw_noise, self.num_data)
        # anything else gpytorch's API allows (combine_terms=False, a prediction that already carries noise): the definition,
        # term by term
        ll = self.likelihood.expected_log_prob(target, pred) / b
        kl = pred.kl / self.num_data
        return ll - kl if self.combine_terms else (ll, kl)
