"""CPU: self-checks of the GP equations of record (oracle fp64).  GP parity is UNPINNED against
the reference (no gpytorch available) — these closed-form identities are what pins the oracle."""
import math

import numpy as np
import torch

from oracle import dvg_oracle as orc
from oracle import params


def test_prior_recovery_and_zero_kl_at_init():
    sd, lik = params.gp_state(7, D=6, M=12, trained=False)
    orc.gp_prior_init(sd)
    h = params.normal(8, 9, 6, scale=0.7)
    s, ell, c = orc.gp_hypers(sd)
    tr = orc.gp_predict(h, sd, training=True)
    # at the prior-init S' = K^-1 and m = c: q(f) equals the prior p(f) and KL(q(u)||p(u)) = 0
    np.testing.assert_allclose(tr["kl"].numpy(), 0.0, atol=1e-3)  # chol stored in fp32
    np.testing.assert_allclose(tr["mean"].numpy(), c.double().view(-1, 1).expand(-1, 9).numpy(), atol=1e-6)
    np.testing.assert_allclose(tr["var"].numpy(), s.double().view(-1, 1).expand(-1, 9).numpy(), rtol=2e-4)
    ev = orc.gp_predict(h, sd, training=False)
    x = h.double().t()
    np.testing.assert_allclose(ev["cov"].numpy(), orc.rbf(x, x, s.double(), ell.double()).numpy(), atol=2e-4)


def test_eval_covariance_is_psd_and_consistent_with_train_variance():
    sd, lik = params.gp_state(11, D=5, M=10)
    h = params.normal(12, 16, 5, scale=0.8)
    ev = orc.gp_predict(h, sd, training=False)
    tr = orc.gp_predict(h, sd, training=True)
    eig = torch.linalg.eigvalsh(ev["cov"])
    assert float(eig.min()) > -1e-9
    np.testing.assert_allclose(ev["mean"].numpy(), tr["mean"].numpy(), atol=1e-12)
    # train variance = eval diagonal wherever the clamp is inactive
    np.testing.assert_allclose(tr["var"].numpy(), torch.diagonal(ev["cov"], dim1=1, dim2=2).numpy(), atol=1e-9)
    assert float(tr["kl"].min()) >= 0.0


def test_kl_matches_generic_gaussian_kl():
    """KL(N(m, K S' K) || N(c, K)) computed from first principles equals the whitened closed form."""
    sd, _ = params.gp_state(21, D=4, M=8)
    h = params.normal(22, 5, 4)
    tr = orc.gp_predict(h, sd, training=True)
    s, ell, c = [t.double() for t in orc.gp_hypers(sd)]
    z = sd["variational_strategy.inducing_points"].squeeze(-1).double()
    K = orc.rbf(z, z, s, ell) + orc.GP_JITTER * torch.eye(8, dtype=torch.float64)
    Ls = torch.tril(sd["variational_strategy.variational_distribution.chol_variational_covar"].double())
    m = sd["variational_strategy.variational_distribution.variational_mean"].double()
    for d in range(4):
        q = torch.distributions.MultivariateNormal(m[d], K[d] @ Ls[d] @ Ls[d].t() @ K[d])
        # the reference parameterisation: q(u) has covariance K S' K and mean m, prior N(c, K);
        # its KL reduces to the closed form when (m-c) is measured in the K^-1 metric
        p = torch.distributions.MultivariateNormal(c[d].expand(8), K[d])
        want = torch.distributions.kl_divergence(q, p)
        np.testing.assert_allclose(float(tr["kl"][d]), float(want), rtol=1e-8, atol=1e-8)


def test_rsample_covariance():
    sd, lik = params.gp_state(31, D=3, M=8)
    h = params.normal(32, 6, 3)
    noise = orc.likelihood_noise(lik)
    ev = orc.gp_predict(h, sd, training=False, noise=noise)
    Lc = torch.linalg.cholesky(ev["cov"])
    np.testing.assert_allclose((Lc @ Lc.transpose(1, 2)).numpy(), ev["cov"].numpy(), atol=1e-12)
    rng = np.random.default_rng(0)
    eps = torch.from_numpy(rng.normal(size=(20000, 3, 6)))
    smp = torch.stack([orc.gp_rsample(ev["mean"], ev["cov"], e) for e in eps[:4000]])
    emp = torch.einsum("ndi,ndj->dij", smp - ev["mean"], smp - ev["mean"]) / smp.shape[0]
    assert float((emp - ev["cov"]).abs().max()) < 0.15 * float(ev["cov"].abs().max())
    # noise enters the diagonal only
    ev0 = orc.gp_predict(h, sd, training=False)
    d = ev["cov"] - ev0["cov"]
    np.testing.assert_allclose(d.numpy(), (noise.double().view(-1, 1, 1) * torch.eye(6, dtype=torch.float64)).numpy(),
                               atol=1e-12)


def test_elbo_definition():
    sd, lik = params.gp_state(41, D=4, M=8)
    h = params.normal(42, 7, 4)
    tgt = params.normal(43, 4, 7)
    noise = orc.likelihood_noise(lik)
    tr = orc.gp_predict(h, sd, training=True)
    elbo = orc.variational_elbo(tr, tgt, noise, num_data=7)
    nz = noise.double().view(-1, 1)
    manual = (-0.5 * ((tgt.double() - tr["mean"]) ** 2 + tr["var"]) / nz - 0.5 * nz.log()
              - 0.5 * math.log(2 * math.pi)).mean(-1) - tr["kl"] / 7
    np.testing.assert_allclose(elbo.numpy(), manual.numpy(), rtol=1e-12)
    assert elbo.shape == (4,)


# --------------------------------------------------------------------------------------------------------------------
# Literature anchors.  With the OPTIMAL variational distribution (Titsias 2009, "Variational Learning of Inducing
# Variables in Sparse Gaussian Processes", eq. 10:  q(u) = N(c + s^-2 Kzz Sig Kzx (y - c),  Kzz Sig Kzz),
# Sig = (Kzz + s^-2 Kzx Kxz)^-1) the equations of record must reproduce published closed forms:
#   * the predictive distribution = the DTC / projected-process predictive (Quinonero-Candela & Rasmussen 2005, eq. 20;
#     Titsias 2009 eq. 6), and with Z = X exact GP regression (Rasmussen & Williams 2006, eq. 2.25 / 2.26);
#   * B x ELBO = the collapsed bound  log N(y | c, Qnn + s^2 I) - tr(Knn - Qnn) / (2 s^2)  (Titsias 2009 eq. 9), and
#     with Z = X the exact log marginal likelihood (R&W eq. 2.30).
# In this repository's parameterisation cov(u) = Kzz S' Kzz, i.e. S' = Sig, L_S = chol(Sig), m = E[u].
# These pin the composition of mean / covariance / KL / expected log-likelihood / ELBO scaling against results that do
# not come from this repository (the numbers gpytorch itself would give remain unpinned: it is not installable here).
# --------------------------------------------------------------------------------------------------------------------
def _optimal_q(sd, lik, x, y, jitter):
    """Installs Titsias' optimal q(u) for data (x (D,N), y (D,N)) into sd (fp64); returns the pieces the checks need."""
    s, ell, c = [t.double() for t in orc.gp_hypers(sd)]
    nz = orc.likelihood_noise(lik).double().reshape(-1)
    z = sd["variational_strategy.inducing_points"].squeeze(-1).double()
    D, M = z.shape
    kzz = orc.rbf(z, z, s, ell) + jitter * torch.eye(M, dtype=torch.float64)
    kzx = orc.rbf(z, x, s, ell)
    sig = torch.linalg.inv(kzz + kzx @ kzx.transpose(1, 2) / nz.view(-1, 1, 1))
    r = (y - c.view(-1, 1)).unsqueeze(-1)
    sd["variational_strategy.variational_distribution.variational_mean"] = \
        c.view(-1, 1) + (kzz @ sig @ kzx @ r).squeeze(-1) / nz.view(-1, 1)
    sd["variational_strategy.variational_distribution.chol_variational_covar"] = torch.linalg.cholesky(sig)
    return s, ell, c, nz, z, kzz, kzx, sig, r


def test_optimal_q_gives_the_dtc_predictive_and_the_collapsed_bound(monkeypatch):
    jitter = 1e-9
    monkeypatch.setattr(orc, "GP_JITTER", jitter)
    D, M, N, NS = 3, 7, 12, 5
    sd, lik = params.gp_state(61, D=D, M=M)
    sd["variational_strategy.inducing_points"] = torch.linspace(-1.0, 1.0, M).repeat(D, 1).unsqueeze(-1) + \
        0.05 * params.normal(62, D, M, 1)
    x = params.normal(63, D, N, scale=0.6).double()
    y = params.normal(64, D, N, scale=0.8).double()
    s, ell, c, nz, z, kzz, kzx, sig, r = _optimal_q(sd, lik, x, y, jitter)
    # predictive at new points, eval mode: DTC mean and covariance
    xs = params.normal(65, D, NS, scale=0.7).double()
    ev = orc.gp_predict(xs.t().contiguous(), sd, training=False)
    ksz = orc.rbf(xs, z, s, ell)
    mean = c.view(-1, 1) + (ksz @ sig @ kzx @ r).squeeze(-1) / nz.view(-1, 1)
    cov = orc.rbf(xs, xs, s, ell) - ksz @ torch.linalg.inv(kzz) @ ksz.transpose(1, 2) + ksz @ sig @ ksz.transpose(1, 2)
    np.testing.assert_allclose(ev["mean"].numpy(), mean.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ev["cov"].numpy(), cov.numpy(), rtol=1e-6, atol=1e-7)
    # ELBO on the training data: B x (ll / B - KL / num_data) with num_data = B is Titsias' collapsed bound
    tr = orc.gp_predict(x.t().contiguous(), sd, training=True)
    elbo = orc.variational_elbo(tr, y, nz, num_data=N) * N
    qnn = kzx.transpose(1, 2) @ torch.linalg.inv(kzz) @ kzx
    knn_diag = s.view(-1, 1).expand(-1, N)
    for d in range(D):
        mvn = torch.distributions.MultivariateNormal(c[d].expand(N), qnn[d] + nz[d] * torch.eye(N, dtype=torch.float64))
        bound = mvn.log_prob(y[d]) - 0.5 * (knn_diag[d] - torch.diagonal(qnn[d])).sum() / nz[d]
        np.testing.assert_allclose(float(elbo[d]), float(bound), rtol=1e-6, atol=1e-6)


def test_inducing_points_at_the_data_recover_exact_gp_regression(monkeypatch):
    jitter = 1e-10
    monkeypatch.setattr(orc, "GP_JITTER", jitter)
    D, N, NS = 3, 8, 6
    sd, lik = params.gp_state(71, D=D, M=N)
    x = (torch.linspace(-1.0, 1.0, N).repeat(D, 1) + 0.08 * params.normal(72, D, N)).double()
    sd["variational_strategy.inducing_points"] = x.unsqueeze(-1).clone()          # Z = X
    y = params.normal(73, D, N, scale=0.8).double()
    s, ell, c, nz, z, kzz, kzx, sig, r = _optimal_q(sd, lik, x, y, jitter)
    xs = params.normal(74, D, NS, scale=0.7).double()
    ev = orc.gp_predict(xs.t().contiguous(), sd, training=False)
    kxx = orc.rbf(x, x, s, ell)
    ksx = orc.rbf(xs, x, s, ell)
    inv = torch.linalg.inv(kxx + nz.view(-1, 1, 1) * torch.eye(N, dtype=torch.float64))
    mean = c.view(-1, 1) + (ksx @ inv @ r).squeeze(-1)                             # R&W eq. 2.25 (constant mean c)
    cov = orc.rbf(xs, xs, s, ell) - ksx @ inv @ ksx.transpose(1, 2)                # R&W eq. 2.26
    np.testing.assert_allclose(ev["mean"].numpy(), mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ev["cov"].numpy(), cov.numpy(), rtol=1e-5, atol=1e-6)
    # likelihood(.) adds the noise to the diagonal: the predictive of y*, R&W eq. 2.24 + sigma^2 I
    evn = orc.gp_predict(xs.t().contiguous(), sd, training=False, noise=nz)
    np.testing.assert_allclose(evn["cov"].numpy(), (cov + nz.view(-1, 1, 1) * torch.eye(NS, dtype=torch.float64)).numpy(),
                               rtol=1e-5, atol=1e-6)
    # and the ELBO is tight: B x ELBO = log N(y | c, K + sigma^2 I)  (R&W eq. 2.30)
    tr = orc.gp_predict(x.t().contiguous(), sd, training=True)
    elbo = orc.variational_elbo(tr, y, nz, num_data=N) * N
    for d in range(D):
        mvn = torch.distributions.MultivariateNormal(c[d].expand(N), kxx[d] + nz[d] * torch.eye(N, dtype=torch.float64))
        np.testing.assert_allclose(float(elbo[d]), float(mvn.log_prob(y[d])), rtol=1e-5, atol=1e-5)
