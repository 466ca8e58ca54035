"""Winograd batched GEMM and direct 3x3 conv per vgg_64 layer shape, back to back at steady clocks (GPU only): the A/B
workload for builds of the library (DVG_HIP_LIB=..., tools/ab_variants.sh, make f32mfma) and the DVG_GEMM_* switches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops
from dvg_amd._lib import lib
from tools.bench_small import time_fn
dev = torch.device("cuda:0"); p = ops._p; s = ops._stream
out = []
for (N, H, C, Cout) in [(64, 32, 128, 128), (64, 16, 256, 256), (64, 8, 512, 512), (576, 16, 256, 256)]:
    T = N * (H // 4) ** 2
    v = torch.randn((36, T, C), device=dev); m = torch.empty((36, T, Cout), device=dev)
    u = ops.winograd_weight(torch.randn(Cout, C, 3, 3, device=dev) * 0.02, 4)
    t = time_fn(lambda: lib().dvg_gemm_batched_k16(p(v), p(u), p(m), 36, T // 16, 16, C, Cout, s()), iters=100)
    out.append(f"{H}^2 {C}->{Cout} B{N}: {t:6.1f} us {2e-6 * 36 * T * C * Cout / t:5.1f} TF")
print(" | ".join(out))
out = []
for (N, H, C, Cout) in [(64, 64, 64, 64), (64, 32, 128, 128), (64, 16, 256, 256)]:
    x = ops.nhwc_empty(N, C, H, H, dev).normal_()
    w = torch.randn(Cout, C, 3, 3, device=dev) * 0.02
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    wp = ops.pack_igemm_weight(w)
    t = time_fn(lambda: ops.conv3x3(x, None, wp, sc, sh), iters=50)
    out.append(f"conv3 {H}^2 {C}->{Cout}: {t:6.1f} us {2e-6 * N * H * H * Cout * 9 * C / t:5.1f} TF")
print(" | ".join(out))
