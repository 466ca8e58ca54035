#!/usr/bin/env python3
"""Direct (dvg_conv_wgrad_multi) vs Winograd-form (dvg_winograd_wgrad_*) weight gradient of the 3x3 layers, 8 uses per
launch as the training path batches them (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from tools.bench_layers import time_fn  # noqa: E402

SHAPES = [(64, 32, 128, 128), (64, 16, 128, 256), (64, 16, 256, 256), (64, 16, 256, 128), (64, 8, 256, 512), (64, 8, 512, 512),
          (64, 8, 512, 256), (16, 8, 512, 512), (16, 16, 256, 256)]


def main():
    dev = torch.device("cuda:0")
    items = int(os.environ.get("BENCH_ITEMS", "8"))
    for (N, H, C, Cout) in SHAPES:
        xs = [ops.nhwc_empty(N, C, H, H, dev).normal_() for _ in range(items)]
        dus = [ops.nhwc_empty(N, Cout, H, H, dev).normal_() for _ in range(items)]
        fl = items * 2.0 * N * H * H * Cout * 9 * C
        td = time_fn(lambda: ops.conv_wgrad_partial_multi(ops.MODE_CONV3, xs, None, dus))
        tw = time_fn(lambda: ops.winograd_wgrad_partial_multi(xs, dus))
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        for _ in range(3):
            ops.winograd_wgrad_partial_multi(xs, dus)
        ops.set_timer(None)
        parts = {k: round(1e3 * v["ms"] / 3, 1) for k, v in timer.summary().items()}
        g = timer.summary()["winograd_wgrad_gemm"]
        print(f"N {N:3d} {H:2d}x{H:<2d} {C:3d}->{Cout:3d} x{items}: direct {td:7.1f} us ({fl / td / 1e6:5.1f} TF) | winograd {tw:7.1f} us "
              f"x{td / tw:.2f}  per call us {parts}  gemm {g['flops'] / g['ms'] / 1e9:5.1f} TF executed")


if __name__ == "__main__":
    main()
