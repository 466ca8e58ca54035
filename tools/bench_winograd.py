#!/usr/bin/env python3
"""Direct implicit GEMM vs the Winograd F(2x2,3x3) path per eval-mode 3x3 layer shape (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from tools.bench_layers import time_fn  # noqa: E402

SHAPES = [(64, 64, 64, 64), (64, 32, 128, 128), (64, 16, 128, 256), (64, 16, 256, 256), (64, 16, 256, 128), (64, 8, 256, 512),
          (64, 8, 512, 512), (64, 8, 512, 256), (576, 16, 256, 256), (576, 8, 512, 512)]


def main():
    dev = torch.device("cuda:0")
    for (N, H, C, Cout) in SHAPES:
        if C < int(os.environ.get("BENCH_MIN_C", "0")):
            continue
        x = ops.nhwc_empty(N, C, H, H, dev).normal_()
        w = torch.randn(Cout, C, 3, 3, device=dev) * 0.02
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        wp = ops.pack_igemm_weight(w)
        td = time_fn(lambda: ops.conv3x3(x, None, wp, sc, sh))
        fl = 2.0 * N * H * H * Cout * 9 * C
        line = f"N {N:3d} {H:2d}x{H:<2d} {C:3d}->{Cout:3d}: direct {td:7.1f} us ({fl / td / 1e6:5.1f} TF)"
        for m in (2, 4):
            if not ops.winograd_ok(N, C, H, H, Cout, m):
                continue
            u = ops.winograd_weight(w, m)
            tw = time_fn(lambda: ops.conv3x3_winograd(x, u, sc, sh))
            timer = ops.KernelTimer()
            ops.set_timer(timer)
            for _ in range(5):
                ops.conv3x3_winograd(x, u, sc, sh)
            ops.set_timer(None)
            parts = [round(1e3 * v["ms"] / v["launches"], 1) for v in timer.summary().values()]
            line += f" | F{m}: {tw:7.1f} us x{td / tw:.2f} {parts}"
        print(line)


if __name__ == "__main__":
    main()
