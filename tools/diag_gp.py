#!/usr/bin/env python3
"""Phase stamps of gp_predict_kernel (debug hook dvg_debug_set_gp_clockbuf): cycles per phase, B=64 eval + sample."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops
from dvg_amd._lib import LIB_PATH
from oracle import params
lib = ctypes.CDLL(LIB_PATH); lib.dvg_debug_set_gp_clockbuf.argtypes = [ctypes.c_void_p, ctypes.c_uint]
dev = torch.device("cuda:0"); B, D, M = 64, 90, 40
gsd, lsd = params.gp_state(3, D, M)
g = {k: v.to(dev) for k, v in gsd.items()}
h = torch.tanh(torch.randn(B, D, device=dev)); eps = torch.randn(D, B, device=dev)
import torch.nn.functional as F
s = F.softplus(g["covar_module.raw_outputscale"]).reshape(-1); ell = F.softplus(g["covar_module.base_kernel.raw_lengthscale"]).reshape(-1)
c = g["mean_module.constant"].reshape(-1); noise = F.softplus(lsd["noise_covar.raw_noise"].to(dev)).reshape(-1) + 1e-4
run = lambda: ops.gp_predict(h, g["variational_strategy.inducing_points"], g["variational_strategy.variational_distribution.variational_mean"],
                             g["variational_strategy.variational_distribution.chol_variational_covar"], c, s, ell, noise=noise, eps=eps, jitter=1e-3)
for _ in range(20): run()
torch.cuda.synchronize()
buf = torch.zeros(D * 12, dtype=torch.int64, device=dev)
lib.dvg_debug_set_gp_clockbuf(ctypes.c_void_p(buf.data_ptr()), buf.numel() // 12); run(); torch.cuda.synchronize(); lib.dvg_debug_set_gp_clockbuf(ctypes.c_void_p(0), 0)
d = buf.cpu().numpy().reshape(D, 12).astype(np.float64)
names = ["assemble", "chol(Kzz) || W", "fwd subst", "mean/var(+kl)", "covariance", "chol(Sigma)", "sample"]
for i, n in enumerate(names):
    print(f"{n:18s} {np.mean(d[:, i + 1] - d[:, i]):9.0f} cycles")
print(f"{'total':18s} {np.mean(d[:, 7] - d[:, 0]):9.0f} cycles = {np.mean(d[:, 7] - d[:, 0]) / 2.37e3:.1f} us at 2.37 GHz")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize(); print("per call (eager, incl. launch)", e0.elapsed_time(e1) / 50 * 1e3, "us")
