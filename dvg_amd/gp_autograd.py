"""Train-mode GP with gradients (train.py:164-169,225-232: the ELBO and the decoded GP mean both
back-propagate into the GP parameters AND into the encoder through h).

Forward: ONE `dvg_gp_predict` launch (mean, clamped marginal variance, KL).
Backward: ONE `dvg_gp_train_bwd` launch producing the gradients w.r.t. h, the inducing points, the
variational mean / Cholesky factor and the (soft-plus'ed) hyper-parameters; the soft-plus chain and
the noise term live in ordinary autograd ops on 90-element vectors.
"""
from __future__ import annotations

import torch

from . import ops


class _GPTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, z, m, ls, c, s, ell, jitter):
        r = ops.gp_predict(h, z, m, ls, c, s, ell, want_var=True, want_kl=True, train_mode=True, jitter=jitter)
        ctx.save_for_backward(h, z, m, ls, c, s, ell)
        ctx.jitter = jitter
        return r["mean"], r["var"], r["kl"]

    @staticmethod
    def backward(ctx, dmean, dvar, dkl):
        h, z, m, ls, c, s, ell = ctx.saved_tensors
        g = ops.gp_train_bwd(h, z, m, ls, c, s, ell, dmean, dvar, dkl, ctx.jitter)
        return g["dh"], g["dz"].view_as(z), g["dm"], g["dls"], g["dc"].view_as(c), g["ds"].view_as(s), \
            g["dell"].view_as(ell), None


STEP_GROUPS = True   # several time steps of a latent dim per workgroup where dvg_gp_step_group says so (tests switch it off to
                     # compare with one workgroup per (step, dim))


class _GPTrainSteps(torch.autograd.Function):
    """_GPTrain for S time steps side by side: h (B, S * D), ONE parameter set of D latent dims read with period D by the
    kernels (no tiled parameter copies).  The workgroups take one step or - small batches - a group of steps each
    (ops.gp_step_group); backward sums the per-workgroup parameter gradients over the groups in one launch."""

    @staticmethod
    def forward(ctx, h, z, m, ls, c, s, ell, jitter, S):
        D = z.shape[0]
        k = ops.gp_step_group(h.shape[0], S, D, z.shape[1]) if STEP_GROUPS else 1
        r = ops.gp_predict(h, z, m, ls, c, s, ell, want_var=True, want_kl=True, train_mode=True, jitter=jitter, param_period=D,
                           step_group=k)
        ctx.save_for_backward(h, z, m, ls, c, s, ell)
        ctx.jitter, ctx.S, ctx.k = jitter, S, k
        return r["mean"], r["var"], r["kl"]

    @staticmethod
    def backward(ctx, dmean, dvar, dkl):
        h, z, m, ls, c, s, ell = ctx.saved_tensors
        g = ops.gp_train_bwd(h, z, m, ls, c, s, ell, dmean, dvar, dkl, ctx.jitter, param_period=z.shape[0], step_group=ctx.k)
        grads = [g["dz"], g["dm"], g["dls"], g["dc"], g["ds"], g["dell"]]
        dz, dm, dls, dc, ds, dell = ops.sum_steps(grads, g["groups"]) if g["groups"] > 1 else grads
        return g["dh"], dz.view_as(z), dm.view_as(m), dls.view_as(ls), dc.view_as(c), ds.view_as(s), dell.view_as(ell), None, None


class _GPElboSteps(torch.autograd.Function):
    """_GPElbo on S x D rows with ONE raw-noise vector of D entries (noise period D)."""

    @staticmethod
    def forward(ctx, mean, var, kl, target, raw_noise, num_data, S):
        ctx.save_for_backward(mean, var, kl, target, raw_noise)
        ctx.num_data, ctx.S = num_data, S
        return ops.gp_elbo(mean, var, kl, target, raw_noise, num_data, noise_period=raw_noise.numel())

    @staticmethod
    def backward(ctx, gelbo):
        mean, var, kl, target, raw_noise = ctx.saved_tensors
        gm, gv, gk, gt, gr = ops.gp_elbo_bwd(mean, var, kl, target, raw_noise, gelbo, ctx.num_data,
                                             need_gtarget=ctx.needs_input_grad[3], noise_period=raw_noise.numel())
        (gr,) = ops.sum_steps([gr], ctx.S)
        return gm, gv, gk.view_as(kl), gt, gr.view_as(raw_noise), None, None


def gp_train(layer, h, noise):
    from .models.gp_models import JITTER
    vs = layer.variational_strategy
    s, ell, c = layer.hypers()
    mean, var, kl = _GPTrain.apply(h if h.is_contiguous() else h.contiguous(), vs.inducing_points,
                                   vs.variational_distribution.variational_mean,
                                   vs.variational_distribution.chol_variational_covar, c, s, ell, JITTER)
    if noise is not None:
        var = var + noise.view(-1, 1)
    return {"mean": mean, "var": var, "kl": kl, "sample": None, "cov": None}


def gp_elbo_steps(layer, mll, hin, htgt):
    """The GP posterior + ELBO term of S teacher-forced time steps in ONE forward and ONE backward launch each
    (train.py:164-169 and :225-226 call `mll(gp_layer(h_i), h_target_i)` once per step; the steps share the GP's parameters
    and do not depend on each other).  hin, htgt: (S, B, D).  The S x D (step, latent dim) pairs are laid out as S * D
    virtual latent dims - the kernels are one workgroup per dim - whose workgroups read the ONE parameter set with period D
    (r05; until r04 the parameters were tiled S times with torch `repeat`: 7 copies forward, 7 reshape + sum launches
    backward per call) and whose parameter gradients are summed over the steps by one launch; the steps' codes sit side by
    side in a (B, S * D) matrix.  Returns (elbo (S * D,), mean (S, B, D)): `-elbo.sum()` is the closure's sum over the steps of
    `-mll(...).sum()`, mean[i] what `gp_layer(h_i).mean.transpose(0, 1)` would be.  Values per (step, dim) are exactly
    the per-step calls' (same kernels, same arithmetic)."""
    from .models.gp_models import JITTER
    layer.ensure_initialized()
    S, B, D = hin.shape
    vs = layer.variational_strategy
    vd = vs.variational_distribution
    s, ell, c = layer.hypers()
    hp = hin.permute(1, 0, 2).reshape(B, S * D)
    mean, var, kl = _GPTrainSteps.apply(hp, vs.inducing_points.squeeze(-1), vd.variational_mean, vd.chol_variational_covar,
                                        c, s, ell, JITTER, S)
    tgt = htgt.permute(1, 0, 2).reshape(B, S * D).transpose(0, 1)          # (S * D, B), strided like h_target.transpose(0, 1)
    elbo = _GPElboSteps.apply(mean, var, kl, tgt, mll.likelihood.noise_covar.raw_noise.reshape(-1), mll.num_data, S)
    return elbo, mean.view(S, D, B).transpose(1, 2)


class _GPElbo(torch.autograd.Function):
    """VariationalELBO(combine_terms=True)(pred, target) as one launch forward and one backward (dvg_gp_elbo /
    dvg_gp_elbo_bwd) instead of ~25 torch launches on (D,B) tensors per GP call: 950 of a dcgan_64 iteration's launches."""

    @staticmethod
    def forward(ctx, mean, var, kl, target, raw_noise, num_data):
        ctx.save_for_backward(mean, var, kl, target, raw_noise)
        ctx.num_data = num_data
        return ops.gp_elbo(mean, var, kl, target, raw_noise, num_data)

    @staticmethod
    def backward(ctx, gelbo):
        mean, var, kl, target, raw_noise = ctx.saved_tensors
        gm, gv, gk, gt, gr = ops.gp_elbo_bwd(mean, var, kl, target, raw_noise, gelbo, ctx.num_data,
                                             need_gtarget=ctx.needs_input_grad[3])
        return gm, gv, gk.view_as(kl), gt, gr.view_as(raw_noise), None


def gp_elbo(mean, var, kl, target, raw_noise, num_data):
    return _GPElbo.apply(mean, var, kl, target, raw_noise, num_data)
