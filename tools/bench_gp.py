#!/usr/bin/env python3
"""Per-call time of the GP kernels at training shapes (train-mode predict + backward), steady clocks (GPU only)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from oracle import params  # noqa: E402
from tools.bench_small import time_fn  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    D, M = 90, 40
    gsd, _ = params.gp_state(3, D, M)
    g = {k: v.to(dev) for k, v in gsd.items()}
    z = g["variational_strategy.inducing_points"]
    m = g["variational_strategy.variational_distribution.variational_mean"]
    ls = g["variational_strategy.variational_distribution.chol_variational_covar"]
    s = F.softplus(g["covar_module.raw_outputscale"]).reshape(-1)
    ell = F.softplus(g["covar_module.base_kernel.raw_lengthscale"]).reshape(-1)
    c = g["mean_module.constant"].reshape(-1)
    for B in (16, 64, 128):
        h = torch.tanh(torch.randn(B, D, device=dev))
        gm, gv, gk = torch.randn(D, B, device=dev), torch.randn(D, B, device=dev), torch.randn(D, device=dev)
        bwd = time_fn(lambda: ops.gp_train_bwd(h, z, m, ls, c, s, ell, gm, gv, gk, 1e-3), iters=100)
        fwd = time_fn(lambda: ops.gp_predict(h, z, m, ls, c, s, ell, train_mode=True, want_kl=True, jitter=1e-3), iters=100)
        eps = torch.randn(D, B, device=dev)
        smp = time_fn(lambda: ops.gp_predict(h, z, m, ls, c, s, ell, noise=s, eps=eps, jitter=1e-3), iters=100)
        print(f"B={B:4d}  gp_train_bwd {bwd:7.1f} us   gp_predict(train, KL) {fwd:7.1f} us   gp_predict(eval, sample) {smp:7.1f} us")
    # the S time steps of a training closure side by side (gp_autograd.gp_elbo_steps): S * D workgroups on one parameter set
    for B, S in ((16, 11), (4, 15), (64, 19)):
        h = torch.tanh(torch.randn(B, S * D, device=dev))
        gm, gv, gk = torch.randn(S * D, B, device=dev), torch.randn(S * D, B, device=dev), torch.randn(S * D, device=dev)
        bwd = time_fn(lambda: ops.gp_train_bwd(h, z, m, ls, c, s, ell, gm, gv, gk, 1e-3, param_period=D), iters=50)
        fwd = time_fn(lambda: ops.gp_predict(h, z, m, ls, c, s, ell, train_mode=True, want_kl=True, jitter=1e-3, param_period=D), iters=50)
        print(f"B={B:4d} S={S:3d} ({S * D} workgroups)  gp_train_bwd {bwd:7.1f} us   gp_predict(train, KL) {fwd:7.1f} us")
        # groups of k steps per workgroup (dvg_gp_step_group's choice is marked *)
        auto = ops.gp_step_group(B, S, D, M)
        for k in sorted({2, 3, 4, 6, 8, S, auto} - {1}):
            if k > S:
                continue
            try:
                bwd = time_fn(lambda: ops.gp_train_bwd(h, z, m, ls, c, s, ell, gm, gv, gk, 1e-3, param_period=D, step_group=k), iters=50)
                fwd = time_fn(lambda: ops.gp_predict(h, z, m, ls, c, s, ell, train_mode=True, want_kl=True, jitter=1e-3,
                                                     param_period=D, step_group=k), iters=50)
            except RuntimeError as e:      # the group's points do not fit the LDS
                print(f"    k={k:2d}: {str(e)[:90]}")
                continue
            print(f"    k={k:2d}{'*' if k == auto else ' '} ({-(-S // k) * D:4d} workgroups)  gp_train_bwd {bwd:7.1f} us   "
                  f"gp_predict(train, KL) {fwd:7.1f} us")


if __name__ == "__main__":
    main()
