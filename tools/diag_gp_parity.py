#!/usr/bin/env python3
"""Achieved error of the GP kernels against the fp64 oracle, beside the oracle's own fp32-vs-fp64 deviation (the
yardstick of tests/test_gpu_parity.py::test_gp_predict_eval_and_train).  GPU only; prints one line per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from dvg_amd._lib import lib  # noqa: E402
from oracle import dvg_oracle as orc, params  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ad = lambda a, b: float((a.double().cpu() - b.double()).abs().max())  # noqa: E731
    for B, D, M in [(64, 90, 40), (50, 90, 40), (16, 12, 40), (95, 6, 40), (128, 8, 40), (7, 5, 64)]:
        sd, lik = params.gp_state(90, D=D, M=M)
        h = params.normal(91, B, D, scale=0.7).tanh()
        noise = orc.likelihood_noise(lik)
        s, ell, c = orc.gp_hypers(sd)
        eps = params.normal(92, D, B)
        args = [sd["variational_strategy.inducing_points"], sd["variational_strategy.variational_distribution.variational_mean"],
                sd["variational_strategy.variational_distribution.chol_variational_covar"], c, s, ell]
        args = [t.to(dev) for t in args]
        ref = orc.gp_predict(h, sd, False, noise)
        f32 = orc.gp_predict(h, sd, False, noise, dtype=torch.float32)
        r = ops.gp_predict(h.to(dev), *args, noise=noise.to(dev), eps=eps.to(dev), want_cov=True)
        smp = orc.gp_rsample(ref["mean"], ref["cov"], eps.double())
        sc, ms = float(ref["cov"].abs().max()), float(ref["mean"].abs().max())
        tr = orc.gp_predict(h, sd, True)
        tf = orc.gp_predict(h, sd, True, dtype=torch.float32)
        t = ops.gp_predict(h.to(dev), *args, want_kl=True, train_mode=True)
        rk = lambda a: float(((a.double().cpu() - tr["kl"]).abs() / tr["kl"].abs()).max())  # noqa: E731
        print(f"B={B:3d} D={D:2d} M={M} prec {lib().dvg_gp_precision(B, M, 1)}/{lib().dvg_gp_precision(B, M, 0)} | "
              f"mean {ad(r['mean'], ref['mean']) / ms:.1e} (oracle-f32 {ad(f32['mean'], ref['mean']) / ms:.1e})  "
              f"cov {ad(r['cov'], ref['cov']) / sc:.1e} ({ad(f32['cov'], ref['cov']) / sc:.1e})  "
              f"sample {ad(r['sample'], smp) / float(smp.abs().max()):.1e} | train var {ad(t['var'], tr['var']) / sc:.1e} "
              f"({ad(tf['var'], tr['var']) / sc:.1e})  kl {rk(t['kl']):.1e} ({rk(tf['kl']):.1e})")


if __name__ == "__main__":
    main()
