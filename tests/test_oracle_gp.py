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


# --------------------------------------------------------------------------------------------------------------------
# r06: what can be hardened on a row whose reference arithmetic (gpytorch) is not installable
# --------------------------------------------------------------------------------------------------------------------
def test_second_independent_oracle_agrees_in_the_trained_regime():
    """oracle/gp_unwhitened.py states the same sparse variational GP in the textbook un-whitened form (numpy, solve /
    slogdet, no shared code); for trained states - S' != K_zz^-1, where the whitened and un-whitened algebra differ in every
    term - mean, full covariance (with and without likelihood noise), train-mode variance and KL agree to 1e-10 in fp64."""
    from oracle import gp_unwhitened as uw
    for seed, (D, M, B) in ((101, (5, 12, 9)), (102, (7, 40, 16)), (103, (3, 8, 33))):
        sd, lik = params.gp_state(seed, D=D, M=M, trained=True)
        sd = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
        h = params.normal(seed + 50, B, D, scale=0.7).double()
        noise = orc.likelihood_noise(lik).double()
        ev = orc.gp_predict(h, sd, training=False, noise=noise)
        tr = orc.gp_predict(h, sd, training=True, noise=noise)
        ev0 = orc.gp_predict(h, sd, training=False)
        other = uw.predict(h.numpy(), {k: v.numpy() for k, v in sd.items()}, orc.GP_JITTER, noise=noise.numpy())
        other0 = uw.predict(h.numpy(), {k: v.numpy() for k, v in sd.items()}, orc.GP_JITTER)
        scale = float(ev["cov"].abs().max())
        np.testing.assert_allclose(ev["mean"].numpy(), other["mean"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(ev["cov"].numpy(), other["cov"], rtol=0, atol=1e-10 * max(1.0, scale))
        np.testing.assert_allclose(ev0["cov"].numpy(), other0["cov"], rtol=0, atol=1e-10 * max(1.0, scale))
        np.testing.assert_allclose(tr["var"].numpy(), other["var_train"], rtol=0, atol=1e-10 * max(1.0, scale))
        np.testing.assert_allclose(tr["kl"].numpy(), other["kl"], rtol=1e-10, atol=1e-9)
        # the trained regime really is one: S' is far from K_zz^-1 (else this test would only re-check the prior)
        assert float(tr["kl"].min()) > 1.0


def test_first_call_initialisation_both_jitter_readings():
    """gp_models.INIT_JITTER_TERMS (the admitted open point of DESIGN.md 3.3): 1 -> L_S = chol((K_zz + 1e-3 I)^-1), the GP starts
    exactly at its prior (KL = 0, variance = outputscale); 2 -> chol((K_zz + 2e-3 I)^-1), the reading recalled for the later
    gpytorch 0.3.x releases: KL small but positive, variance within 2e-3 of the prior's.  The module (pure torch here) and the
    oracle agree for both, and a TRAINED state (variational_params_initialized = 1) is left untouched by either."""
    from dvg_amd.models import gp_models as gm
    from oracle import gp_unwhitened as uw
    D, M = 6, 12
    old = gm.INIT_JITTER_TERMS
    try:
        kls = {}
        for terms in (1, 2):
            gm.INIT_JITTER_TERMS = terms
            sd, _ = params.gp_state(7, D=D, M=M, trained=False)
            layer = gm.GPRegressionLayer1(D, M)
            layer.load_state_dict(sd)
            assert int(layer.variational_strategy.variational_params_initialized) == 0
            layer.ensure_initialized()
            assert int(layer.variational_strategy.variational_params_initialized) == 1
            ref = {k: v.clone() for k, v in sd.items()}
            orc.gp_prior_init(ref, jitter_terms=terms)
            got = layer.state_dict()
            for k in ("variational_strategy.variational_distribution.chol_variational_covar",
                      "variational_strategy.variational_distribution.variational_mean"):
                np.testing.assert_allclose(got[k].numpy(), ref[k].numpy(), rtol=1e-6, atol=1e-7)
            h = params.normal(8, 9, D, scale=0.7)
            tr = orc.gp_predict(h, ref, training=True)
            s = orc.gp_hypers(ref)[0].double()
            kls[terms] = float(tr["kl"].max())
            dev = float((tr["var"] - s.view(-1, 1)).abs().max())
            if terms == 1:
                assert kls[1] < 1e-3 and dev < 2e-4 * float(s.max())      # chol stored in fp32
            else:
                # starts NEAR the prior, not at it.  Closed form: in K_zz's eigenbasis S' K = diag(r_i), r_i = (lam_i + j) /
                # (lam_i + 2 j), so KL = 1/2 sum_i [r_i - 1 - ln r_i] - up to 0.097 nats per near-null direction of K_zz (an
                # RBF Gram matrix of 12-40 points has many); the predictive variance moves by a few per cent of the outputscale
                z = ref["variational_strategy.inducing_points"].squeeze(-1).double()
                ell = orc.gp_hypers(ref)[1].double()
                lam = torch.linalg.eigvalsh(orc.rbf(z, z, s, ell))
                r = (lam + orc.GP_JITTER) / (lam + 2 * orc.GP_JITTER)
                want = 0.5 * (r - 1 - torch.log(r)).sum(1)
                np.testing.assert_allclose(tr["kl"].numpy(), want.numpy(), rtol=2e-3, atol=2e-4)     # L_S stored in fp32
                assert 1e-3 < kls[2] < 0.097 * M and 1e-4 < dev < 0.05 * float(s.max()), (kls, dev)
                chk = uw.predict(h.numpy(), {k: v.double().numpy() for k, v in ref.items()}, orc.GP_JITTER)
                np.testing.assert_allclose(tr["kl"].numpy(), chk["kl"], rtol=1e-6, atol=1e-8)
            # a trained state: the flag is set, nothing is re-initialised, whatever the constant says
            sdt, _ = params.gp_state(9, D=D, M=M, trained=True)
            trained = gm.GPRegressionLayer1(D, M)
            trained.load_state_dict(sdt)
            trained.ensure_initialized()
            for k, v in trained.state_dict().items():
                assert torch.equal(v, sdt[k]), k
        assert kls[2] > 10 * kls[1]
    finally:
        gm.INIT_JITTER_TERMS = old


def test_state_dict_loader_accepts_and_reports_gpytorch_variants(capsys):
    """generate_frames.py:67-72 of the reference loads real gpytorch-0.3.x state_dicts.  Ours are strict on the key set but
    accept - and report, never silently drop - the shape variants of each key ((90,1,1) / (90,1) / (90,) ...), the constraint
    buffers gpytorch >= 0.3.3 stores (checked against the bounds the kernels hard-wire: different bounds raise) and a
    missing initialisation flag beside trained variational parameters; an unknown key still fails the strict load."""
    import pytest
    from dvg_amd.models import gp_models as gm
    D, M = 5, 8
    sd, lik = params.gp_state(13, D=D, M=M, trained=True)
    variant = dict(sd)
    variant["mean_module.constant"] = sd["mean_module.constant"].reshape(D)                      # (D,)    for (D,1)
    variant["covar_module.raw_outputscale"] = sd["covar_module.raw_outputscale"].reshape(D, 1)   # (D,1)   for (D,)
    variant["covar_module.base_kernel.raw_lengthscale"] = sd["covar_module.base_kernel.raw_lengthscale"].reshape(D, 1)
    variant["variational_strategy.variational_distribution.variational_mean"] = \
        sd["variational_strategy.variational_distribution.variational_mean"].reshape(D, 1, M)
    variant["covar_module.raw_outputscale_constraint.lower_bound"] = torch.tensor(0.0)
    variant["covar_module.raw_outputscale_constraint.upper_bound"] = torch.tensor(float("inf"))
    variant["covar_module.base_kernel.raw_lengthscale_constraint.lower_bound"] = torch.tensor(0.0)
    variant["covar_module.base_kernel.raw_lengthscale_constraint.upper_bound"] = torch.tensor(float("inf"))
    del variant["variational_strategy.variational_params_initialized"]
    layer = gm.GPRegressionLayer1(D, M)
    layer.load_state_dict(variant)
    rep = layer.load_report
    assert len(rep["reshaped"]) == 4 and len(rep["constraints"]) == 4 and len(rep["assumed"]) == 1
    err = capsys.readouterr().err
    assert "reshaped" in err and "raw_lengthscale_constraint.lower_bound" in err and "variational_params_initialized" in err
    plain = gm.GPRegressionLayer1(D, M)
    plain.load_state_dict(sd)
    assert not any(plain.load_report.values())
    for (k, a), (_, b) in zip(layer.state_dict().items(), plain.state_dict().items()):
        assert torch.equal(a, b), k
    likv = dict(lik)
    likv["noise_covar.raw_noise"] = lik["noise_covar.raw_noise"].reshape(D)
    likv["noise_covar.raw_noise_constraint.lower_bound"] = torch.tensor(1e-4)
    likv["noise_covar.raw_noise_constraint.upper_bound"] = torch.tensor(float("inf"))
    like = gm.GaussianLikelihood(batch_size=D)
    like.load_state_dict(likv)
    assert torch.equal(like.noise_covar.raw_noise, lik["noise_covar.raw_noise"]) and len(like.load_report["constraints"]) == 2
    # different bounds = different arithmetic: refused
    bad = dict(likv)
    bad["noise_covar.raw_noise_constraint.lower_bound"] = torch.tensor(1e-6)
    with pytest.raises(RuntimeError, match="hard-wires"):
        gm.GaussianLikelihood(batch_size=D).load_state_dict(bad)
    # unknown keys / wrong element counts still fail the strict load
    with pytest.raises(RuntimeError):
        gm.GPRegressionLayer1(D, M).load_state_dict(dict(sd, **{"covar_module.period_length": torch.zeros(D)}))
    with pytest.raises(RuntimeError):
        gm.GPRegressionLayer1(D, M).load_state_dict(dict(sd, **{"mean_module.constant": torch.zeros(D + 1, 1)}))
