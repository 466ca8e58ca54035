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
