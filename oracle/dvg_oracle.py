"""CPU ORACLE for the DVG frame-prediction hot path.  *** TEST INFRASTRUCTURE ONLY ***

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
this module, and only as the checker / reported baseline.  The product (dvg_amd/)
never imports it and has no CPU fallback.

It restates, with plain torch-CPU functional ops on explicit state_dicts (reference
key names), what the reference's modules compute.  Each function cites the reference
lines it follows (paths relative to the reference tree).

Pinning:
  * encoder / decoder / LSTM / gaussian_encoder restatements are pinned against golden vectors produced
    by the reference's own modules imported on CPU (tests/golden/make_golden.py,
    committed with its outputs; checked by tests/test_oracle_golden.py) - forward outputs, BatchNorm side
    effects AND the gradients the reference's own `.backward()` produces (B=16 train-mode encoder->decoder,
    4-step LSTM BPTT), against which torch-autograd-of-this-restatement is checked.
  * step closures / rollouts (train_model_loss, plot_rollout, posterior_rollout, gp_trigger_gen, best_of_n_sse,
    best_ssim): restated from the reference SOURCE (module-level script code, not importable); their integer
    bookkeeping is pinned by that text, their arithmetic by the pinned blocks they call (+ the GP caveat below).
  * GP (gp_models.py + gpytorch): **parity unpinned** — gpytorch is not vendored in
    the reference, not pinned by any manifest and not installable here.  The GP
    functions below restate the gpytorch 0.3.x WhitenedVariationalStrategy /
    GaussianLikelihood / VariationalELBO arithmetic as documented in DESIGN.md
    ("GP: equations of record") and are self-checked in fp64 against closed-form
    identities and against PUBLISHED results they must reproduce with the optimal q(u) of Titsias 2009: the DTC
    predictive, the collapsed bound, and - inducing points at the data - exact GP regression and its log marginal
    likelihood (Rasmussen & Williams eq. 2.25 / 2.26 / 2.30)  (tests/test_oracle_gp.py).
  * evaluation metrics (utils.eval_seq -> skimage compare_ssim / compare_psnr): **parity unpinned** as well -
    scikit-image is neither vendored, pinned nor installed; `ssim_skimage` / `psnr_skimage` restate its published
    algorithm with the defaults the reference relies on, pinned by closed forms and a brute-force window evaluation
    (tests/test_oracle_golden.py::test_eval_metrics_oracle_identities).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def _bn(x: torch.Tensor, sd: SD, prefix: str, training: bool) -> torch.Tensor:
    """nn.BatchNorm2d (vgg_64.py:9, dcgan_64.py:9): batch statistics in train mode
    (running stats updated in place, momentum 0.1), running statistics in eval mode."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)


# Forced kinks (gradient parity only).  LeakyReLU and MaxPool2d are the two non-smooth points of the backbones: an element whose
# pre-activation is within rounding of 0 (or a 2x2 window whose two largest entries are within rounding of each other) takes
# a different branch in two implementations that agree to 1e-7 in the forward, and from there their GRADIENTS differ by O(1)
# of that element's contribution.  Inside `with forced_kinks(acts):` the oracle takes its branch decisions from recorded
# activations of the implementation under test (acts[prefix] = that layer's OUTPUT, (N,C,H,W)): LeakyReLU's slope from the
# sign of the recorded output, the max-pool winner as the arg-max of the recorded output (first maximum in scan order, the
# nn.MaxPool2d rule).  Everything else - every convolution, BatchNorm statistic, sum - stays the oracle's own arithmetic, so
# the comparison then isolates the ARITHMETIC of the backward pass from the choice of branches.
_FORCED = None
_RECORD = None


class record_kinks:
    """`with record_kinks() as acts:` fills acts[prefix] with every LeakyReLU layer's output (what forced_kinks consumes)."""

    def __enter__(self):
        global _RECORD
        self.prev, _RECORD = _RECORD, {}
        return _RECORD

    def __exit__(self, *exc):
        global _RECORD
        _RECORD = self.prev


class forced_kinks:
    def __init__(self, acts: Dict[str, torch.Tensor]):
        self.acts = acts

    def __enter__(self):
        global _FORCED
        self.prev, _FORCED = _FORCED, self.acts
        return self

    def __exit__(self, *exc):
        global _FORCED
        _FORCED = self.prev


def _lrelu(z: torch.Tensor, name: str) -> torch.Tensor:
    """LeakyReLU(0.2) (vgg_64.py:10, dcgan_64.py:10,22); under forced_kinks the branch follows the recorded output's sign."""
    if _FORCED is not None and name in _FORCED:
        one = torch.ones((), dtype=z.dtype)
        slope = torch.where(_FORCED[name] > 0, one, 0.2 * one)      # 0.2 rounded in z's dtype, like F.leaky_relu
        out = z * slope
    else:
        out = F.leaky_relu(z, 0.2)
    if _RECORD is not None:
        _RECORD[name] = out.detach()
    return out


def _maxpool(h: torch.Tensor, name: str) -> torch.Tensor:
    """MaxPool2d(2,2) (vgg_64.py:49); under forced_kinks the winner of each window is the recorded output's arg-max."""
    if _FORCED is not None and name in _FORCED:
        _, idx = F.max_pool2d(_FORCED[name], 2, 2, return_indices=True)
        return h.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    return F.max_pool2d(h, 2, 2)


def vgg_layer(x: torch.Tensor, sd: SD, prefix: str, training: bool) -> torch.Tensor:
    """vgg_64.py:5-15: Conv2d(nin,nout,3,1,1) -> BatchNorm2d -> LeakyReLU(0.2)."""
    y = F.conv2d(x, sd[prefix + ".main.0.weight"], sd[prefix + ".main.0.bias"], stride=1, padding=1)
    return _lrelu(_bn(y, sd, prefix + ".main.1", training), prefix)


def dcgan_conv(x: torch.Tensor, sd: SD, prefix: str, training: bool) -> torch.Tensor:
    """dcgan_64.py:4-14: Conv2d(nin,nout,4,2,1) -> BatchNorm2d -> LeakyReLU(0.2)."""
    y = F.conv2d(x, sd[prefix + ".main.0.weight"], sd[prefix + ".main.0.bias"], stride=2, padding=1)
    return _lrelu(_bn(y, sd, prefix + ".main.1", training), prefix)


def dcgan_upconv(x: torch.Tensor, sd: SD, prefix: str, training: bool) -> torch.Tensor:
    """dcgan_64.py:16-26: ConvTranspose2d(nin,nout,4,2,1) -> BatchNorm2d -> LeakyReLU(0.2)."""
    y = F.conv_transpose2d(x, sd[prefix + ".main.0.weight"], sd[prefix + ".main.0.bias"], stride=2, padding=1)
    return _lrelu(_bn(y, sd, prefix + ".main.1", training), prefix)


def _count_blocks(sd: SD, stage: str) -> int:
    n = 0
    while f"{stage}.{n}.main.0.weight" in sd:
        n += 1
    return n


# --------------------------------------------------------------------------------------
# vgg encoder / decoder (64 and 128)
# --------------------------------------------------------------------------------------
def vgg_encoder(x: torch.Tensor, sd: SD, training: bool = False) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """vgg_64.py:51-57 (4 stages + c5 head) and vgg_128.py:56-63 (5 stages + c6 head):
    stage -> skip, MaxPool2d(2,2) between stages, head = Conv2d(512,dim,4,1,0)+BN+Tanh."""
    nstage = 4 if "c6.0.weight" not in sd else 5
    head = f"c{nstage + 1}"
    skips = []
    h = x
    last = None
    for s in range(1, nstage + 1):
        if s > 1:
            h = _maxpool(h, last)
        for b in range(_count_blocks(sd, f"c{s}")):
            last = f"c{s}.{b}"
            h = vgg_layer(h, sd, last, training)
        skips.append(h)
    h = _maxpool(h, last)
    h = F.conv2d(h, sd[head + ".0.weight"], sd[head + ".0.bias"])
    h = torch.tanh(_bn(h, sd, head + ".1", training))
    return h.reshape(-1, sd[head + ".0.weight"].shape[0]), skips


def vgg_decoder(vec: torch.Tensor, skips: Sequence[torch.Tensor], sd: SD, training: bool = False) -> torch.Tensor:
    """vgg_64.py:95-106 / vgg_128.py:107-120: upc1 = ConvTranspose2d(dim,512,4,1,0)+BN+LReLU;
    then [nearest x2 -> cat(skip) -> vgg blocks] per stage; the last stage ends with
    ConvTranspose2d(64,nc,3,1,1)+Sigmoid."""
    nstage = len(skips)
    dim = sd["upc1.0.weight"].shape[0]
    d = F.conv_transpose2d(vec.reshape(-1, dim, 1, 1), sd["upc1.0.weight"], sd["upc1.0.bias"])
    d = _lrelu(_bn(d, sd, "upc1.1", training), "upc1")
    for s in range(nstage):
        stage = f"upc{s + 2}"
        d = torch.cat([F.interpolate(d, scale_factor=2, mode="nearest"), skips[nstage - 1 - s]], 1)
        for b in range(_count_blocks(sd, stage)):
            d = vgg_layer(d, sd, f"{stage}.{b}", training)
    last = f"upc{nstage + 1}"
    nb = _count_blocks(sd, last)
    d = F.conv_transpose2d(d, sd[f"{last}.{nb}.weight"], sd[f"{last}.{nb}.bias"], stride=1, padding=1)
    return torch.sigmoid(d)


def vgg_gaussian_encoder(x: torch.Tensor, sd: SD, eps: torch.Tensor, training: bool = False):
    """vgg_64.gaussian_encoder.forward (vgg_64.py:155-165): the encoder trunk, then mu / logvar heads on h5 and
    z = eps * exp(0.5 * logvar) + mu (:150-153) with the N(0,1) draw passed in.  Returns (z, mu, logvar, skips)."""
    h, skips = vgg_encoder(x, sd, training)
    mu = F.linear(h, sd["mu_net.weight"], sd["mu_net.bias"])
    logvar = F.linear(h, sd["logvar_net.weight"], sd["logvar_net.bias"])
    return eps * torch.exp(0.5 * logvar) + mu, mu, logvar, skips


# --------------------------------------------------------------------------------------
# dcgan encoder / decoder (64 and 128)
# --------------------------------------------------------------------------------------
def dcgan_encoder(x: torch.Tensor, sd: SD, training: bool = False) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """dcgan_64.py:48-54 / dcgan_128.py:50-57: strided conv stages, every stage output is
    a skip; head Conv2d(512,dim,4,1,0)+BN+Tanh."""
    nstage = 4 if "c6.0.weight" not in sd else 5
    head = f"c{nstage + 1}"
    skips = []
    h = x
    for s in range(1, nstage + 1):
        h = dcgan_conv(h, sd, f"c{s}", training)
        skips.append(h)
    h = F.conv2d(h, sd[head + ".0.weight"], sd[head + ".0.bias"])
    h = torch.tanh(_bn(h, sd, head + ".1", training))
    return h.reshape(-1, sd[head + ".0.weight"].shape[0]), skips


def dcgan_decoder(vec: torch.Tensor, skips: Sequence[torch.Tensor], sd: SD, training: bool = False,
                  final_act: str = "tanh") -> torch.Tensor:
    """dcgan_64.py:81-88 (final Tanh, :75-79) / dcgan_128.py:86-94 (final Sigmoid, :80-84):
    upc1 stem, then dcgan_upconv on cat([d, skip]) per stage, last layer
    ConvTranspose2d(128,nc,4,2,1) + activation."""
    nstage = len(skips)
    dim = sd["upc1.0.weight"].shape[0]
    d = F.conv_transpose2d(vec.reshape(-1, dim, 1, 1), sd["upc1.0.weight"], sd["upc1.0.bias"])
    d = _lrelu(_bn(d, sd, "upc1.1", training), "upc1")
    for s in range(nstage - 1):
        d = dcgan_upconv(torch.cat([d, skips[nstage - 1 - s]], 1), sd, f"upc{s + 2}", training)
    last = f"upc{nstage + 1}"
    d = F.conv_transpose2d(torch.cat([d, skips[0]], 1), sd[last + ".0.weight"], sd[last + ".0.bias"], stride=2,
                           padding=1)
    return torch.tanh(d) if final_act == "tanh" else torch.sigmoid(d)


# --------------------------------------------------------------------------------------
# lstm
# --------------------------------------------------------------------------------------
def lstm_init_hidden(batch: int, hidden: int, n_layers: int, dtype=torch.float32):
    """lstm.py:58-63: zeros (h, c) per layer."""
    return [(torch.zeros(batch, hidden, dtype=dtype), torch.zeros(batch, hidden, dtype=dtype))
            for _ in range(n_layers)]


def lstm_cell(x, hc, sd: SD, prefix: str):
    """nn.LSTMCell (lstm.py:51,69), gate order i,f,g,o."""
    h, c = hc
    g = F.linear(x, sd[prefix + ".weight_ih"], sd[prefix + ".bias_ih"]) + \
        F.linear(h, sd[prefix + ".weight_hh"], sd[prefix + ".bias_hh"])
    i, f, gg, o = g.chunk(4, 1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def lstm_step(x: torch.Tensor, sd: SD, hidden: list) -> torch.Tensor:
    """lstm.py:65-72: embed -> n_layers LSTMCells (state mutated in `hidden`) -> Linear+Tanh."""
    in_size = sd["embed.weight"].shape[1]
    h_in = F.linear(x.reshape(-1, in_size), sd["embed.weight"], sd["embed.bias"])
    for i in range(len(hidden)):
        hidden[i] = lstm_cell(h_in, hidden[i], sd, f"lstm.{i}")
        h_in = hidden[i][0]
    return torch.tanh(F.linear(h_in, sd["output.0.weight"], sd["output.0.bias"]))


def gaussian_lstm_step(x: torch.Tensor, sd: SD, hidden: list, eps: torch.Tensor):
    """lstm.py:166-175 with the N(0,1) draw of :163 passed in as `eps`."""
    in_size = sd["embed.weight"].shape[1]
    h_in = F.linear(x.reshape(-1, in_size), sd["embed.weight"], sd["embed.bias"])
    for i in range(len(hidden)):
        hidden[i] = lstm_cell(h_in, hidden[i], sd, f"lstm.{i}")
        h_in = hidden[i][0]
    mu = F.linear(h_in, sd["mu_net.weight"], sd["mu_net.bias"])
    logvar = F.linear(h_in, sd["logvar_net.weight"], sd["logvar_net.bias"])
    return eps * torch.exp(0.5 * logvar) + mu, mu, logvar


# --------------------------------------------------------------------------------------
# GP: equations of record.  State keys follow gpytorch 0.3.x:
#   variational_strategy.inducing_points (D,M,1)
#   variational_strategy.variational_distribution.variational_mean (D,M)
#   variational_strategy.variational_distribution.chol_variational_covar (D,M,M)
#   mean_module.constant (D,1), covar_module.raw_outputscale (D),
#   covar_module.base_kernel.raw_lengthscale (D,1,1); likelihood: noise_covar.raw_noise (D,1)
# --------------------------------------------------------------------------------------
GP_JITTER = 1e-3
NOISE_FLOOR = 1e-4


def softplus(x):
    return F.softplus(x)


def gp_hypers(sd: SD):
    s = softplus(sd["covar_module.raw_outputscale"]).reshape(-1)
    ell = softplus(sd["covar_module.base_kernel.raw_lengthscale"]).reshape(-1)
    c = sd["mean_module.constant"].reshape(-1)
    return s, ell, c


def likelihood_noise(lsd: SD):
    """GaussianLikelihood(batch_size=D) (train.py:102): softplus(raw_noise) + 1e-4 floor."""
    return softplus(lsd["noise_covar.raw_noise"]).reshape(-1) + NOISE_FLOOR


def rbf(a: torch.Tensor, b: torch.Tensor, s: torch.Tensor, ell: torch.Tensor) -> torch.Tensor:
    """ScaleKernel(RBFKernel) (gp_models.py:17-19): a (D,Na), b (D,Nb) -> (D,Na,Nb)."""
    diff = a.unsqueeze(-1) - b.unsqueeze(-2)
    return s.view(-1, 1, 1) * torch.exp(-0.5 * diff * diff / (ell.view(-1, 1, 1) ** 2))


def gp_prior_init(sd: SD, jitter_terms: int = 1) -> None:
    """WhitenedVariationalStrategy.initialize_variational_dist (first call):
    m <- prior mean, L_S <- chol((K_ZZ + jitter_terms * jitter I)^-1) computed in fp64.  jitter_terms: 1 = the GP starts exactly
    at its prior; 2 = the behaviour recalled for the later gpytorch 0.3.x releases, whose initialisation adds the default
    jitter to a prior covariance that already carries it (dvg_amd/models/gp_models.py: INIT_JITTER_TERMS)."""
    s, ell, c = gp_hypers(sd)
    z = sd["variational_strategy.inducing_points"].squeeze(-1)
    M = z.shape[1]
    kzz = rbf(z.double(), z.double(), s.double(), ell.double()) + jitter_terms * GP_JITTER * torch.eye(M, dtype=torch.float64)
    ls = torch.linalg.cholesky(torch.linalg.inv(kzz))
    sd["variational_strategy.variational_distribution.variational_mean"] = c.view(-1, 1).expand(-1, M).clone().to(
        z.dtype)
    sd["variational_strategy.variational_distribution.chol_variational_covar"] = ls.to(z.dtype)


def gp_predict(h: torch.Tensor, sd: SD, training: bool, noise: torch.Tensor = None, dtype=torch.float64) -> dict:
    """gp_layer(h.transpose(0,1).view(D,B,1)) (train.py:225; generate_frames.py:170) and,
    with `noise`, likelihood(gp_layer(...)) (generate_frames.py:131,170).
    h: (B,D).  Returns mean (D,B), var (D,B), cov (D,B,B) [eval], kl (D) [train]."""
    s, ell, c = [t.to(dtype) for t in gp_hypers(sd)]
    z = sd["variational_strategy.inducing_points"].squeeze(-1).to(dtype)
    m = sd["variational_strategy.variational_distribution.variational_mean"].to(dtype)
    ls = torch.tril(sd["variational_strategy.variational_distribution.chol_variational_covar"].to(dtype))
    x = h.to(dtype).t().contiguous()  # (D,B): GP d sees latent column d
    D, M = z.shape
    kzz = rbf(z, z, s, ell) + GP_JITTER * torch.eye(M, dtype=dtype)
    L = torch.linalg.cholesky(kzz)
    kzx = rbf(z, x, s, ell)
    A = torch.linalg.solve_triangular(L, kzx, upper=False)
    v = torch.linalg.solve_triangular(L, (m - c.view(-1, 1)).unsqueeze(-1), upper=False)
    mean = c.view(-1, 1) + (A.transpose(1, 2) @ v).squeeze(-1)
    W = ls.transpose(1, 2) @ kzx
    out = {"mean": mean}
    nz = torch.zeros(D, dtype=dtype) if noise is None else noise.to(dtype).reshape(-1)
    if training:
        var = (W * W).sum(1) + torch.clamp(s.view(-1, 1) - (A * A).sum(1), min=0.0)
        out["var"] = var + nz.view(-1, 1)
        logdet_k = 2.0 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(1)
        logdet_s = 2.0 * torch.log(torch.abs(torch.diagonal(ls, dim1=1, dim2=2))).sum(1)
        trace = ((ls @ ls.transpose(1, 2)) * kzz).sum((1, 2))
        quad = (v.squeeze(-1) ** 2).sum(1)
        out["kl"] = 0.5 * (-logdet_k - logdet_s + trace + quad - M)
    else:
        cov = W.transpose(1, 2) @ W + rbf(x, x, s, ell) - A.transpose(1, 2) @ A
        cov = cov + nz.view(-1, 1, 1) * torch.eye(x.shape[1], dtype=dtype)
        out["cov"] = cov
        out["var"] = torch.diagonal(cov, dim1=1, dim2=2)
    return out


def gp_rsample(mean: torch.Tensor, cov: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """MultivariateNormal.rsample with the base sample passed in: mean + chol(cov) eps."""
    Lc = torch.linalg.cholesky(cov)
    return mean + (Lc @ eps.to(cov.dtype).unsqueeze(-1)).squeeze(-1)


def variational_elbo(pred: dict, target: torch.Tensor, noise: torch.Tensor, num_data: int) -> torch.Tensor:
    """VariationalELBO(likelihood, gp, num_data, combine_terms=True) (train.py:112,226):
    (1/B) sum_n E_q[log N(y_n | f_n, sigma^2)] - KL/num_data, per latent dim -> (D,).
    `pred` is the TRAIN-mode latent prediction WITHOUT likelihood noise;
    target (D,B) = h_target.transpose(0,1)."""
    mean, var = pred["mean"], pred["var"]
    nz = noise.to(mean.dtype).view(-1, 1)
    ll = -0.5 * ((target.to(mean.dtype) - mean) ** 2 + var) / nz - 0.5 * torch.log(nz) - 0.5 * math.log(2 * math.pi)
    return ll.sum(-1) / mean.shape[-1] - pred["kl"] / num_data


# --------------------------------------------------------------------------------------
# latent / index bookkeeping and the rollout
# --------------------------------------------------------------------------------------
def normalize_data(sequence: torch.Tensor) -> List[torch.Tensor]:
    """utils.py:86-95: (B,T,H,W,C) -> list of T tensors (B,C,H,W)."""
    seq = sequence.transpose(0, 1).transpose(3, 4).transpose(2, 3)
    return [seq[t].contiguous() for t in range(seq.shape[0])]


def gp_trigger_steps(n_past: int, n_eval: int, period: int = 15) -> List[int]:
    """generate_frames.py:165-171: GP sampling replaces the LSTM prediction at steps
    i >= n_past with i % 15 == 0."""
    return [i for i in range(n_past, n_eval) if i % period == 0]


def rollout(x: Sequence[torch.Tensor], enc, dec, lstm_sd: SD, gp_sd: SD, lik_sd: SD, n_past: int, n_eval: int,
            eps_by_step: Dict[int, torch.Tensor], last_frame_skip: bool = False, rnn_size: int = 256,
            n_layers: int = 2, period: int = 15) -> List[torch.Tensor]:
    """One sample of the make_gifs loop (generate_frames.py:143-177), all modules in eval
    mode.  enc(x)->(h,skips), dec(vec,skips)->frame are closures over the encoder/decoder
    restatements.  Returns the n_eval frames [x0, ...] (conditioning frames copied)."""
    B = x[0].shape[0]
    hidden = lstm_init_hidden(B, rnn_size, n_layers, dtype=x[0].dtype)
    frames = [x[0]]
    x_in = x[0]
    skip = None
    noise = likelihood_noise(lik_sd)
    for i in range(1, n_eval):
        h, sk = enc(x_in)
        if last_frame_skip or i < n_past:
            skip = sk
        if i < n_past:
            lstm_step(h, lstm_sd, hidden)  # output discarded (generate_frames.py:162)
            x_in = x[i]
        else:
            h_pred = lstm_step(h, lstm_sd, hidden)
            if i % period == 0:
                p = gp_predict(h, gp_sd, training=False, noise=noise, dtype=torch.float64)
                z = gp_rsample(p["mean"], p["cov"], eps_by_step[i]).t().to(h.dtype)
                x_in = dec(z, skip)
            else:
                x_in = dec(h_pred, skip)
        frames.append(x_in)
    return frames


# --------------------------------------------------------------------------------------
# Evaluation metrics of utils.eval_seq (utils.py:220-234)
#
# The reference calls skimage.measure.compare_ssim / compare_psnr (utils.py:13-14).  scikit-image is a third-party
# dependency that is neither vendored in /root/reference nor pinned by any manifest (the `compare_*` names exist up
# to skimage 0.15) and is not installed in this image, so this is a restatement of its PUBLISHED algorithm with the
# defaults the reference relies on - **parity unpinned** for these two functions (no reference golden vectors):
#   compare_ssim(X, Y): float64; win_size 7; uniform_filter means; sample covariance (cov_norm = NP/(NP-1));
#     K1 = 0.01, K2 = 0.03; data_range = dtype range of float images = 2; mean of S over the image cropped by
#     (win_size-1)//2 on every side.
#   compare_psnr(true, test): data_range = 1 if true.min() >= 0 else 2 (float images); 10 log10(R^2 / mse).
# --------------------------------------------------------------------------------------
def ssim_skimage(x, y) -> float:
    import numpy as np
    from scipy.ndimage import uniform_filter
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    win, k1, k2, rng = 7, 0.01, 0.03, 2.0
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def psnr_skimage(true, test) -> float:
    import numpy as np
    true = np.asarray(true, dtype=np.float64)
    test = np.asarray(test, dtype=np.float64)
    rng = 1.0 if true.min() >= 0 else 2.0
    mse = np.mean((true - test) ** 2)
    return float(10.0 * np.log10(rng * rng / mse))


def eval_seq(gt: Sequence[torch.Tensor], pred: Sequence[torch.Tensor]):
    """utils.eval_seq (utils.py:220-234): (ssim, psnr) arrays (B, T), channel-averaged."""
    import numpy as np
    T, bs = len(gt), gt[0].shape[0]
    ssim, psnr = np.zeros((bs, T)), np.zeros((bs, T))
    for i in range(bs):
        for t in range(T):
            nc = gt[t][i].shape[0]
            for c in range(nc):
                ssim[i, t] += ssim_skimage(gt[t][i][c].numpy(), pred[t][i][c].numpy())
                psnr[i, t] += psnr_skimage(gt[t][i][c].numpy(), pred[t][i][c].numpy())
            ssim[i, t] /= nc
            psnr[i, t] /= nc
    return ssim, psnr


# --------------------------------------------------------------------------------------
# Step closures and qualitative / generation rollouts: the orchestration of rows S and K restated
# from the reference's SOURCE (these functions are module-level script code that cannot be imported; what pins
# them is the reference text itself: loop bounds, the skip rule, which tensor feeds the GP, which index is read).
# enc(x)->(h,skips) and dec(vec,skips)->frame are closures over the encoder/decoder restatements above; their BN mode
# is the caller's choice (train.py keeps encoder/decoder in TRAIN mode during plot(), train.py:372-374).
# --------------------------------------------------------------------------------------
def train_model_loss(x, enc, dec, lstm_sd: SD, gp_sd: SD, lik_sd: SD, n_past: int, n_future: int, num_data: int,
                     last_frame_skip: bool = False, rnn_size: int = 256, n_layers: int = 2, gp_dtype=torch.float32):
    """train.py:200-239 up to the loss (no backward / optimiser): returns (loss, mse_latent)."""
    hidden = lstm_init_hidden(x[0].shape[0], rnn_size, n_layers, dtype=x[0].dtype)
    noise = likelihood_noise(lik_sd)
    mse = mse_latent = mse_gp = ae_mse = 0
    max_ll = 0
    skip = None
    for i in range(1, n_past + n_future):
        h, sk = enc(x[i - 1])                                   # :214
        h_target = enc(x[i])[0]                                 # :215 (not detached)
        if last_frame_skip or i < n_past:                       # :217-220
            skip = sk
        h_pred = lstm_step(h, lstm_sd, hidden)                  # :222
        mse_latent = mse_latent + F.mse_loss(h_pred, h_target)  # :223
        gp = gp_predict(h, gp_sd, training=True, dtype=gp_dtype)                                 # :225
        max_ll = max_ll - variational_elbo(gp, h_target.t(), noise, num_data=num_data)           # :226
        x_pred = dec(h_pred, skip)                              # :227
        ae_mse = ae_mse + F.mse_loss(dec(h_target, skip), x[i])  # :229-230
        mse = mse + F.mse_loss(x_pred, x[i])                    # :233
        mse_gp = mse_gp + F.mse_loss(dec(gp["mean"].t().to(x[i].dtype), skip), x[i])            # :232,234
    loss = 1000 * ae_mse + 0.001 * mse + 0.01 * mse_latent + 0.001 * mse_gp + 0.0001 * max_ll.sum()   # :239
    return loss, mse_latent


def train_frame_predictor_loss(x, enc, lstm_sd: SD, n_past: int, n_future: int, last_frame_skip: bool = False,
                               rnn_size: int = 256, n_layers: int = 2):
    """train.py:175-193: sum over steps of MSE(lstm(h_{i-1}), h_i)."""
    hidden = lstm_init_hidden(x[0].shape[0], rnn_size, n_layers, dtype=x[0].dtype)
    mse_latent = 0
    for i in range(1, n_past + n_future):
        h = enc(x[i - 1])[0]
        h_target = enc(x[i])[0]
        mse_latent = mse_latent + F.mse_loss(lstm_step(h, lstm_sd, hidden), h_target)
    return mse_latent


def train_gp_loss(x, enc, gp_sd: SD, lik_sd: SD, n_past: int, n_future: int, num_data: int, gp_dtype=torch.float32):
    """train.py:146-169: sum over steps and latent dims of -ELBO(gp(h_{i-1}), h_i)."""
    noise = likelihood_noise(lik_sd)
    max_ll = 0
    for i in range(1, n_past + n_future):
        h = enc(x[i - 1])[0]
        h_target = enc(x[i])[0].detach()
        gp = gp_predict(h, gp_sd, training=True, dtype=gp_dtype)
        max_ll = max_ll - variational_elbo(gp, h_target.t(), noise, num_data=num_data)
    return max_ll.sum()


def plot_rollout(x, enc, dec, lstm_sd: SD, gp_sd: SD, lik_sd: SD, n_past: int, n_eval: int,
                 eps_by_sample: Sequence[torch.Tensor], last_frame_skip: bool = False, rnn_size: int = 256,
                 n_layers: int = 2, gp_step: int = 10, gp_dtype=torch.float64) -> List[List[torch.Tensor]]:
    """train.py:256-289: nsample = len(eps_by_sample) rollouts; the ONE GP-sampled step is `i == 10` (:281) - not the
    i % 15 schedule of generate_frames.py - and it only exists when 10 >= n_past; the GP is fed the ENCODER output h
    (:283).  On conditioning steps the reference also runs `encoder(x[i])` and discards it (:273-274): a BatchNorm
    side effect when enc is in train mode, reproduced here by calling enc.  Returns gen_seq[s][t]."""
    noise = likelihood_noise(lik_sd)
    gen_seq = []
    for eps in eps_by_sample:
        hidden = lstm_init_hidden(x[0].shape[0], rnn_size, n_layers, dtype=x[0].dtype)      # :263
        seq = [x[0]]
        x_in = x[0]
        skip = None
        for i in range(1, n_eval):
            h, sk = enc(x_in)                                              # :267
            if last_frame_skip or i < n_past:                              # :268-271
                skip = sk
            if i < n_past:
                enc(x[i])                                                  # :274, output discarded
                lstm_step(h, lstm_sd, hidden)                              # :276, output discarded
                x_in = x[i]
            else:
                h_pred = lstm_step(h, lstm_sd, hidden)
                if i == gp_step:                                           # :281
                    p = gp_predict(h, gp_sd, training=False, noise=noise, dtype=gp_dtype)
                    x_in = dec(gp_rsample(p["mean"], p["cov"], eps).t().to(h.dtype), skip)   # :283-284
                else:
                    x_in = dec(h_pred, skip)                               # :288
            seq.append(x_in)
        gen_seq.append(seq)
    return gen_seq


def best_of_n_sse(gt: Sequence[torch.Tensor], gen_seq: Sequence[Sequence[torch.Tensor]], nrow: int) -> List[int]:
    """train.py:303-310: per batch row i < nrow the sample with the smallest sum over t of squared error; strict `<`
    from min_mse = 1e7, so ties keep the FIRST sample."""
    best = []
    for i in range(nrow):
        min_mse, min_idx = 1e7, None
        for s in range(len(gen_seq)):
            mse = 0
            for t in range(len(gen_seq[s])):
                mse = mse + torch.sum((gt[t][i] - gen_seq[s][t][i]) ** 2)
            if mse < min_mse:
                min_mse, min_idx = mse, s
        best.append(min_idx)
    return best


def posterior_rollout(x, enc, dec, lstm_sd: SD, gp_sd: SD, lik_sd: SD, n_past: int, n_eval: int,
                      last_frame_skip: bool = False, rnn_size: int = 256, n_layers: int = 2,
                      gp_dtype=torch.float64) -> List[torch.Tensor]:
    """generate_frames.py:110-134: after the conditioning frames EVERY step decodes the GP predictive MEAN, and the GP is
    fed the LSTM OUTPUT h_pred (:131), unlike the sample rollouts (:170) which feed it the encoder output."""
    hidden = lstm_init_hidden(x[0].shape[0], rnn_size, n_layers, dtype=x[0].dtype)
    frames = [x[0]]
    x_in = x[0]
    skip = None
    for i in range(1, n_eval):
        h, sk = enc(x_in)
        if last_frame_skip or i < n_past:
            skip = sk
        if i < n_past:
            lstm_step(h, lstm_sd, hidden)
            x_in = x[i]
        else:
            h_pred = lstm_step(h, lstm_sd, hidden)
            p = gp_predict(h_pred, gp_sd, training=False, noise=likelihood_noise(lik_sd), dtype=gp_dtype)   # :131
            x_in = dec(p["mean"].t().to(h.dtype), skip)                                                      # :132
        frames.append(x_in)
    return frames


def best_ssim(ssim) -> List[int]:
    """generate_frames.py:188-189,207: per batch row, `np.argsort(np.mean(ssim[i], 1))[-1]`; ssim (B, nsample, T)."""
    import numpy as np
    return [int(np.argsort(np.mean(np.asarray(ssim[i]), 1))[-1]) for i in range(len(ssim))]


def gp_trigger_gen(x, enc, dec, lstm_sd: SD, gp_sd: SD, lik_sd: SD, index: int, eps_by_step: Dict[int, torch.Tensor],
                   warmup: int = 12, total: int = 105, depth: int = 1, skip_steps: int = 5, probe: int = 3,
                   rnn_size: int = 256, n_layers: int = 2, gp_dtype=torch.float64,
                   decisions: Optional[Dict[int, bool]] = None, guard: float = 0.0, memo: Optional[dict] = None) -> dict:
    """ONE pass of the `for index in range(batch_size)` body of GPtrigger_gen (generate_frames.py:249-298) with
    `generation` (:220-224) and `var_value` (:227-232) inlined.  The reference's bookkeeping, kept verbatim:
      * the rollout is autoregressive from x[0] alone (x_in = x_out, :280,297) - no ground-truth frame after the first;
      * the skip tensors are those of loop steps i < 5 (:268-269), whatever n_past is;
      * warm-up: 12 steps; the recorded value is the L2 norm over latent dims of the predictive variance (with
        likelihood noise) of sample `index` (:275);
      * afterwards `var_value` reads sample [3] - NOT [index] (:230) - and slides the 12-long window (:231);
      * threshold = mean + (2 + 0.01*depth) * std of the window, numpy population std (:288); `value > threshold`
        decodes a GP sample of the encoder output (:290-292) WITHOUT stepping the LSTM, otherwise `generation` steps it.
    Values are float32 like the reference's `.cpu().numpy()` arrays; the window statistics use numpy on that dtype.
    decisions / guard (parity tests; the analogue of `forced_kinks`): `value > threshold` is a discontinuity - two
    implementations that agree to 1e-7 on both sides can still branch differently when the margin |value - threshold| /
    |threshold| is of that size, and untrained networks roll out towards a fixed point where it is.  With `decisions`
    ({step: bool}, the branches another implementation took) a step whose OWN margin is below `guard` follows that implementation's
    branch and is reported in `forced`; every other step decides for itself.  All arithmetic stays this function's.
    memo (tests): a dict shared by calls on the SAME inputs / parameters / eps: the rollout is a deterministic function of the
    decisions taken so far, so a step already computed under the same decision prefix (another batch index) is looked up -
    frame, recurrent state, per-sample variance norms - instead of recomputed.
    Returns dict(frames, triggers, values, thresholds, margins, forced)."""
    import numpy as np
    hidden = lstm_init_hidden(x[0].shape[0], rnn_size, n_layers, dtype=x[0].dtype)
    noise = likelihood_noise(lik_sd)

    def generation(x_in, skip):                                            # :220-224
        h = enc(x_in)[0]
        return dec(lstm_step(h, lstm_sd, hidden), skip)

    def var_norms(h):
        p = gp_predict(h, gp_sd, training=False, noise=noise, dtype=gp_dtype)
        return np.linalg.norm(p["var"].to(torch.float32).numpy().transpose(), axis=1), p

    context, values, thresholds, triggers, gen_seq, margins, forced = [], [], [], [], [], [], []
    x_in, skip = x[0], None
    for i in range(warmup):                                                # :266-280
        hit = None if memo is None else memo.get(("warm", i))
        if hit is None:
            h, sk = enc(x_in)
            if i < skip_steps:
                skip = sk
            norms = var_norms(h)[0]
            x_in = generation(x_in, skip)
            if memo is not None:
                memo[("warm", i)] = (norms, x_in, list(hidden), skip)
        else:
            norms, x_in, hidden[:], skip = hit[0], hit[1], list(hit[2]), hit[3]
        value = norms[index]                                               # :275
        context.append(value)
        values.append(float(value))
        gen_seq.append(x_in)
    context = np.array(context)                                            # :283
    for i in range(warmup, total):                                         # :285-297
        pre = None if memo is None else memo.get(("pre", i, tuple(triggers)))
        if pre is None:
            h = enc(x_in)[0]
            norms, p = var_norms(h)
            if memo is not None:
                memo[("pre", i, tuple(triggers))] = (h, norms, p)
        else:
            h, norms, p = pre
        value = norms[probe]                                               # :230 - sample 3, not `index`
        context = np.concatenate([context[1:], [value]])                   # :231
        threshold = np.mean(context) + (2 + 0.01 * depth) * np.std(context)   # :288
        take = bool(value > threshold)
        margins.append(float(abs(value - threshold) / abs(threshold)))
        if decisions is not None and margins[-1] < guard:
            take = bool(decisions[i])
            forced.append(i)
        post = None if memo is None else memo.get(("post", i, tuple(triggers), take))
        if post is not None:
            x_in, hidden[:] = post[0], list(post[1])
        elif take:                                                         # :289-292
            x_in = dec(gp_rsample(p["mean"], p["cov"], eps_by_step[i]).t().to(h.dtype), skip)
        else:
            x_in = dec(lstm_step(h, lstm_sd, hidden), skip)                # :295 generation(): its encoder call is `h` above
        if memo is not None and post is None:
            memo[("post", i, tuple(triggers), take)] = (x_in, list(hidden))
        if take:
            triggers.append(i)
        values.append(float(value))
        thresholds.append(float(threshold))
        gen_seq.append(x_in)
    return {"frames": gen_seq, "triggers": triggers, "values": values, "thresholds": thresholds, "margins": margins,
            "forced": forced}
