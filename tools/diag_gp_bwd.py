#!/usr/bin/env python3
"""Phase stamps of gp_train_bwd_kernel (debug hook dvg_debug_set_gp_clockbuf): cycles per phase at B = 64 / 16."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from dvg_amd._lib import LIB_PATH  # noqa: E402
from oracle import params  # noqa: E402

lib = ctypes.CDLL(LIB_PATH)
lib.dvg_debug_set_gp_clockbuf.argtypes = [ctypes.c_void_p, ctypes.c_uint]
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
names = ["load", "assemble K, Kzx", "W, Kzx gm", "chol(K)", "P fill + forward subst", "backward subst", "alpha, gq",
         "GW, G2, GK, dL_S, dm", "RBF chain rule", "block sums + store"]
for B in (64, 16):
    h = torch.tanh(torch.randn(B, D, device=dev))
    gm, gv, gk = torch.randn(D, B, device=dev), torch.randn(D, B, device=dev), torch.randn(D, device=dev)
    run = lambda: ops.gp_train_bwd(h, z, m, ls, c, s, ell, gm, gv, gk, 1e-3)  # noqa: E731
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    buf = torch.zeros(D * 12, dtype=torch.int64, device=dev)
    lib.dvg_debug_set_gp_clockbuf(ctypes.c_void_p(buf.data_ptr()), D)
    run()
    torch.cuda.synchronize()
    lib.dvg_debug_set_gp_clockbuf(ctypes.c_void_p(0), 0)
    d = buf.cpu().numpy().reshape(D, 12).astype(np.float64)
    print(f"B = {B}")
    for i in range(9):
        print(f"  {names[i + 1]:26s} {np.mean(d[:, i + 1] - d[:, i]):9.0f} cycles")
    tot = np.mean(d[:, 9] - d[:, 0])
    print(f"  {'total':26s} {tot:9.0f} cycles = {tot / 2.37e3:.1f} us at 2.37 GHz")
