"""GP trigger wrappers (gp.hip, misc_kernels.hip): the sparse variational GP predictive / sample / KL, its backward, the
ELBO, and GPtrigger_gen's device-side bookkeeping."""
from __future__ import annotations

import torch

from .._lib import check, lib
from ._core import _dev_f32, _p, _run, _stream


# ----------------------------------------------------------------------------------
# GP
# ----------------------------------------------------------------------------------
def gp_predict(h, z, var_mean, chol_var, mean_const, outputscale, lengthscale, *, noise=None, eps=None,
               want_var=True, want_cov=False, want_kl=False, train_mode=False, jitter=1e-3, raw_hypers=False, param_period=0,
               step_group=1):
    """h [B][D] (any strides); returns dict(mean [D][B], var, sample, cov, kl).  raw_hypers: outputscale / lengthscale /
    noise are the RAW parameters, soft-plus'ed (noise: + 1e-4 floor) inside the kernel.
    param_period = P > 0: h carries D = S x P columns - S time steps side by side - and column d uses the parameters of
    latent dim d % P (the parameter tensors have P rows).  step_group = k > 1: k consecutive steps of a latent dim are one
    workgroup's problem (dvg_hip.h; train-mode outputs only) - same outputs."""
    _dev_f32(h, "gp_predict.h")
    h = h if h.is_contiguous() else h.contiguous()
    b, d = h.shape
    m = z.shape[1]
    dev = h.device
    mean = torch.empty((d, b), device=dev, dtype=torch.float32)
    var = torch.empty((d, b), device=dev, dtype=torch.float32) if want_var else None
    sample = torch.empty((d, b), device=dev, dtype=torch.float32) if eps is not None else None
    cov = torch.empty((d, b, b), device=dev, dtype=torch.float32) if want_cov else None
    kl = torch.empty((d,), device=dev, dtype=torch.float32) if want_kl else None
    if eps is not None:
        eps = eps.contiguous()
        if tuple(eps.shape) != (d, b):
            raise RuntimeError(f"gp_predict: eps must be ({d},{b})")
    args = [t.detach().contiguous().view(-1) for t in (z, var_mean, chol_var, mean_const, outputscale, lengthscale)]
    dp = param_period or d
    if d % dp or args[0].numel() != dp * m or args[2].numel() != dp * m * m or args[3].numel() != dp:
        raise RuntimeError("gp_predict: parameter shapes do not match (D,M) / the parameter period")
    nz = None if noise is None else noise.detach().contiguous().view(-1)
    _run("gp_predict", 0.0, 4.0 * (b * d + d * m * (m + 2) + 3 * d * b), lib().dvg_gp_predict, _p(h),
         *[_p(t) for t in args], _p(nz), _p(eps), _p(mean), _p(var), _p(sample), _p(cov), _p(kl), b, d, m,
         int(train_mode) | (2 if raw_hypers else 0), jitter, int(param_period), int(step_group), _stream())
    return {"mean": mean, "var": var, "sample": sample, "cov": cov, "kl": kl}


def gp_var_norms(var: torch.Tensor) -> torch.Tensor:
    """(B,) L2 norm over the latent dims of a predictive variance (D,B): generate_frames.py:230,275's
    `np.linalg.norm(variance.cpu().numpy().transpose(), axis=1)` without the host round trip (dvg_gp_var_norms)."""
    _dev_f32(var, "gp_var_norms.var")
    var = var.contiguous()
    d, b = var.shape
    out = torch.empty(b, device=var.device, dtype=torch.float32)
    check(lib().dvg_gp_var_norms(_p(var), _p(out), d, b, _stream()), "gp_var_norms")
    return out


def gp_trigger_step(var, col, ctx, coef, flag, values, thresholds, flags, slot) -> None:
    """One decision of GPtrigger_gen's main loop on the device (dvg_gp_trigger_step; generate_frames.py:227-232,285-289): ctx
    (window floats) slides in place, flag (1 int32) = value > threshold, logs at `slot`."""
    _dev_f32(var, "gp_trigger_step.var")
    d, b = var.shape
    if not var.is_contiguous() or ctx.dtype != torch.float32 or flag.dtype != torch.int32 or flags.dtype != torch.int32:
        raise RuntimeError("gp_trigger_step: contiguous (D,B) variance, float32 window, int32 flags expected")
    if not 0 <= slot < min(values.numel(), thresholds.numel(), flags.numel()):
        raise RuntimeError("gp_trigger_step: log slot out of range")
    check(lib().dvg_gp_trigger_step(_p(var), d, b, int(col), _p(ctx), ctx.numel(), float(coef), _p(flag), _p(values),
                                    _p(thresholds), _p(flags), int(slot), _stream()), "gp_trigger_step")


def gp_trigger_replay(values, ctx0, coef):
    """(flags int32 (n,), thresholds (n,)): the decisions another batch index would take on the recorded main-loop values
    from its own initial window (dvg_gp_trigger_replay)."""
    _dev_f32(values, "gp_trigger_replay.values")
    _dev_f32(ctx0, "gp_trigger_replay.ctx0")
    values, ctx0 = values.contiguous(), ctx0.contiguous()
    n = values.numel()
    flags = torch.empty(n, dtype=torch.int32, device=values.device)
    thr = torch.empty(n, dtype=torch.float32, device=values.device)
    check(lib().dvg_gp_trigger_replay(_p(values), n, _p(ctx0), ctx0.numel(), float(coef), _p(flags), _p(thr), _stream()),
          "gp_trigger_replay")
    return flags, thr


def gp_trigger_select(flag, sample_db, h_pred, states_old, states_new):
    """(vec (B,D), [state tensors]) of a GPtrigger_gen step (dvg_gp_trigger_select): the GP sample (D,B), transposed, and the
    OLD recurrent state when the device flag is set, the LSTM output and the NEW state otherwise (generate_frames.py:289-296)."""
    import ctypes as C
    _dev_f32(sample_db, "gp_trigger_select.sample")
    _dev_f32(h_pred, "gp_trigger_select.h_pred")
    d, b = sample_db.shape
    if tuple(h_pred.shape) != (b, d) or not sample_db.is_contiguous() or not h_pred.is_contiguous():
        raise RuntimeError("gp_trigger_select: sample (D,B) and h_pred (B,D), both contiguous, expected")
    n = len(states_old)
    if n != len(states_new) or n > 8 or any(a.shape != c.shape or not a.is_contiguous() or not c.is_contiguous()
                                             or a.numel() != states_old[0].numel() for a, c in zip(states_old, states_new)):
        raise RuntimeError("gp_trigger_select: up to 8 contiguous state tensors of one size, old and new alike")
    vec = torch.empty((b, d), device=h_pred.device, dtype=torch.float32)
    outs = [torch.empty_like(a) for a in states_old]
    arr = lambda ts: (C.c_void_p * max(1, n))(*[t.data_ptr() for t in ts])   # noqa: E731
    check(lib().dvg_gp_trigger_select(_p(flag), _p(sample_db), _p(h_pred), _p(vec), d, b, n,
                                      states_old[0].numel() if n else 0, arr(states_old), arr(states_new), arr(outs), _stream()),
          "gp_trigger_select")
    return vec, outs


def gp_elbo(mean, var, kl, target, raw_noise, num_data, noise_period=0):
    """VariationalELBO(combine_terms=True) with the Gaussian likelihood's expected log-probability -> (D,) (dvg_gp_elbo).
    mean, var (D,B) contiguous, kl (D,), target (D,B) with any strides, raw_noise (D,) or (D,1) - (P,) with noise_period = P:
    row d uses raw_noise[d % P]."""
    for t, n in ((mean, "mean"), (var, "var"), (kl, "kl"), (target, "target"), (raw_noise, "raw_noise")):
        _dev_f32(t, "gp_elbo." + n)
    d, b = mean.shape
    if tuple(var.shape) != (d, b) or tuple(target.shape) != (d, b) or kl.numel() != d or raw_noise.numel() != (noise_period or d) \
            or d % (noise_period or d):
        raise RuntimeError(f"gp_elbo: shapes mean {tuple(mean.shape)} var {tuple(var.shape)} target {tuple(target.shape)}")
    mean, var, kl, raw = mean.contiguous(), var.contiguous(), kl.contiguous(), raw_noise.reshape(-1).contiguous()
    out = torch.empty(d, device=mean.device, dtype=torch.float32)
    check(lib().dvg_gp_elbo(_p(mean), _p(var), _p(kl), _p(target), target.stride(0), target.stride(1), _p(raw), _p(out),
                            b, d, int(num_data), int(noise_period), _stream()), "gp_elbo")
    return out


def gp_elbo_bwd(mean, var, kl, target, raw_noise, gelbo, num_data, need_gtarget=True, noise_period=0):
    """Gradients of gp_elbo w.r.t. mean, var (D,B), kl (D,), target (D,B; None unless asked for), raw_noise (D,: one entry per
    ROW also with a noise period - the caller sums the steps)."""
    d, b = mean.shape
    mean, var, kl, raw = mean.contiguous(), var.contiguous(), kl.contiguous(), raw_noise.reshape(-1).contiguous()
    gelbo = gelbo.contiguous()
    dev = mean.device
    gmean, gvar = torch.empty((d, b), device=dev), torch.empty((d, b), device=dev)
    gkl, graw = torch.empty(d, device=dev), torch.empty(d, device=dev)
    gtarget = torch.empty((d, b), device=dev) if need_gtarget else None
    check(lib().dvg_gp_elbo_bwd(_p(mean), _p(var), _p(kl), _p(target), target.stride(0), target.stride(1), _p(raw),
                                _p(gelbo), _p(gmean), _p(gvar), _p(gkl), _p(gtarget), _p(graw), b, d, int(num_data),
                                int(noise_period), _stream()), "gp_elbo_bwd")
    return gmean, gvar, gkl, gtarget, graw


def sum_steps(tensors, steps):
    """[t.view(steps, -1).sum(0) for t in tensors] as ONE launch (dvg_sum_steps_multi; up to 8 tensors): (steps * n_k,) -> (n_k,)."""
    import ctypes as C
    if not 1 <= len(tensors) <= 8:
        raise RuntimeError("sum_steps: 1..8 tensors")
    src = [t.contiguous() for t in tensors]
    for t in src:
        _dev_f32(t, "sum_steps")
        if t.numel() % steps:
            raise RuntimeError("sum_steps: tensor size is not a multiple of the step count")
    dst = [torch.empty(t.numel() // steps, device=t.device, dtype=torch.float32) for t in src]
    k = len(src)
    check(lib().dvg_sum_steps_multi((C.c_void_p * k)(*[t.data_ptr() for t in src]), (C.c_void_p * k)(*[t.data_ptr() for t in dst]),
                                    (C.c_long * k)(*[t.numel() for t in dst]), k, int(steps), _stream()), "sum_steps")
    return dst


def gp_step_group(b, steps, period, m):
    """Steps per workgroup of a time-batched train-mode GP call (dvg_gp_step_group; 1 = one workgroup per (step, dim))."""
    return lib().dvg_gp_step_group(int(b), int(steps), int(period), int(m))


def gp_train_bwd(h, z, m, ls, c, s, ell, gmean, gvar, gkl, jitter=1e-3, param_period=0, step_group=1):
    """Gradients of the train-mode GP prediction (see dvg_gp_train_bwd); with param_period = P the parameter gradients come
    back per WORKGROUP: G x P rows, G = ceil(S / step_group) groups of the D = S x P columns of h (out["groups"] = G;
    sum_steps adds the G copies up)."""
    h = h if h.is_contiguous() else h.contiguous()
    b, d = h.shape
    mm = z.shape[1]
    dev = h.device
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
    period = param_period or d
    k = max(int(step_group), 1)
    groups = -(-(d // period) // k)
    r = groups * period
    out = {"dh": f(b, d), "dz": f(r, mm), "dm": f(r, mm), "dls": f(r, mm, mm), "dc": f(r), "ds": f(r), "dell": f(r),
           "groups": groups}
    args = [t.detach().contiguous().view(-1) for t in (z, m, ls, c, s, ell)]
    g = [None if t is None else t.contiguous() for t in (gmean, gvar, gkl)]
    check(lib().dvg_gp_train_bwd(_p(h), *[_p(t) for t in args], *[_p(t) for t in g], _p(out["dh"]), _p(out["dz"]),
                                 _p(out["dm"]), _p(out["dls"]), _p(out["dc"]), _p(out["ds"]), _p(out["dell"]), b, d,
                                 mm, jitter, int(param_period), k, _stream()), "gp_train_bwd")
    return out
