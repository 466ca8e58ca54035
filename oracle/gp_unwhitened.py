"""TEST INFRASTRUCTURE (checker only; the product never imports oracle/): a SECOND, independently written statement of the
sparse variational GP of gp_models.py:10-24, in the UN-WHITENED textbook parameterisation and in numpy only - no Cholesky
whitening, no triangular solves, no code shared with oracle/dvg_oracle.py.

oracle.gp_predict states the equations of record in the whitened form gpytorch 0.3.x's WhitenedVariationalStrategy computes
(A = L^-1 K_zx, W = L_S^T K_zx, KL from L, L_S).  The same model written the way Hensman et al. (2013, "Gaussian Processes
for Big Data", eq. 3-4 / Titsias 2009) state it:

    q(u) = N(m, S_u),            S_u = K_zz S' K_zz,  S' = L_S L_S^T          (the variational covariance in u space)
    mean(x)  = c + K_xz K_zz^-1 (m - c)
    cov(x)   = K_xx - K_xz K_zz^-1 K_zx + K_xz K_zz^-1 S_u K_zz^-1 K_zx
    KL       = 1/2 [ tr(K_zz^-1 S_u) + (m - c)^T K_zz^-1 (m - c) - M + log|K_zz| - log|S_u| ]

with K_zz^-1 applied by np.linalg.solve and the determinants by np.linalg.slogdet.  tests/test_oracle_gp.py requires the two
to agree to 1e-10 in fp64 for trained states (S' != K_zz^-1), i.e. the whitening algebra of the oracle - the part gpytorch's
own tests would pin - is checked against the textbook form.  (What gpytorch itself computes stays unpinned: not installable.)"""
import numpy as np


def _softplus(x):
    return np.logaddexp(0.0, x)


def _rbf(a, b, s, ell):
    d = a[:, :, None] - b[:, None, :]
    return s[:, None, None] * np.exp(-0.5 * d * d / (ell[:, None, None] ** 2))


def predict(h, sd, jitter, noise=None):
    """h (B, D) latent codes; sd: gpytorch-0.3.x-keyed state_dict (anything np.asarray takes).  Returns fp64 arrays:
    mean (D, B), cov (D, B, B) [+ noise on the diagonal], var_train (D, B) (the train-mode marginal variance with its clamp),
    kl (D,)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float64)    # noqa: E731
    s = _softplus(g("covar_module.raw_outputscale")).reshape(-1)
    ell = _softplus(g("covar_module.base_kernel.raw_lengthscale")).reshape(-1)
    c = g("mean_module.constant").reshape(-1)
    z = g("variational_strategy.inducing_points")[..., 0]
    m = g("variational_strategy.variational_distribution.variational_mean")
    ls = np.tril(g("variational_strategy.variational_distribution.chol_variational_covar"))
    x = np.asarray(h, dtype=np.float64).T
    D, M = z.shape
    kzz = _rbf(z, z, s, ell) + jitter * np.eye(M)
    kzx = _rbf(z, x, s, ell)
    kxx = _rbf(x, x, s, ell)
    su = kzz @ (ls @ np.transpose(ls, (0, 2, 1))) @ kzz
    mean = np.empty((D, x.shape[1]))
    cov = np.empty((D, x.shape[1], x.shape[1]))
    var_train = np.empty((D, x.shape[1]))
    kl = np.empty(D)
    for d in range(D):
        alpha = np.linalg.solve(kzz[d], m[d] - c[d])
        proj = np.linalg.solve(kzz[d], kzx[d])                       # K_zz^-1 K_zx
        mean[d] = c[d] + kzx[d].T @ alpha
        explained = kzx[d].T @ proj
        learned = proj.T @ su[d] @ proj
        cov[d] = kxx[d] - explained + learned
        var_train[d] = np.diag(learned) + np.maximum(s[d] - np.diag(explained), 0.0)
        ld_k = np.linalg.slogdet(kzz[d])[1]
        ld_s = np.linalg.slogdet(su[d])[1]
        kl[d] = 0.5 * (np.trace(np.linalg.solve(kzz[d], su[d])) + (m[d] - c[d]) @ alpha - M + ld_k - ld_s)
    if noise is not None:
        nz = np.asarray(noise, dtype=np.float64).reshape(-1)
        cov = cov + nz[:, None, None] * np.eye(x.shape[1])
        var_train = var_train + nz[:, None]
    return {"mean": mean, "cov": cov, "var_train": var_train, "kl": kl}
