#!/usr/bin/env python3
"""Per-layer timing of the implicit-GEMM conv kernels at the shapes of the B=64 rollout (GPU only).
Prints one line per layer: shape, tile config hint, µs, TFLOP/s, fraction of the fp32-MFMA peak."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402

PEAK = 157.3

VGG64 = [  # (H, C1, C2, Cout, upsample, pool)
    (64, 64, 0, 64, 0, 1), (32, 64, 0, 128, 0, 0), (32, 128, 0, 128, 0, 1), (16, 128, 0, 256, 0, 0),
    (16, 256, 0, 256, 0, 0), (16, 256, 0, 256, 0, 1), (8, 256, 0, 512, 0, 0), (8, 512, 0, 512, 0, 0),
    (8, 512, 0, 512, 0, 1),
    (8, 512, 512, 512, 1, 0), (8, 512, 0, 512, 0, 0), (8, 512, 0, 256, 0, 0), (16, 256, 256, 256, 1, 0),
    (16, 256, 0, 256, 0, 0), (16, 256, 0, 128, 0, 0), (32, 128, 128, 128, 1, 0), (32, 128, 0, 64, 0, 0),
    (64, 64, 64, 64, 1, 0),
]


def time_fn(fn, iters=50, warm_s=0.5):
    # sustained warm-up: after idling the shader clock needs ~100s of ms of load to reach its steady ~2.36 GHz
    t0 = time.time()
    while time.time() - t0 < warm_s:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # µs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N = args.batch
    tot_us = tot_fl = 0.0
    for (H, C1, C2, Cout, up, pool) in VGG64:
        hx = H // 2 if up else H
        x = ops.nhwc_empty(N, C1, hx, hx, dev).normal_()
        sk = ops.nhwc_empty(N, C2, H, H, dev).normal_() if C2 else None
        wp = ops.pack_igemm_weight(torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.02)
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        us = time_fn(lambda: ops.conv3x3(x, sk, wp, sc, sh, upsample=bool(up), pool=bool(pool)))
        fl = 2.0 * N * H * H * Cout * 9 * (C1 + C2)
        tf = fl / us / 1e6
        tot_us += us
        tot_fl += fl
        print(f"conv3x3 {H:3d}x{H:<3d} Cin {C1 + C2:4d} Cout {Cout:3d} up {up} pool {pool}: {us:8.1f} us  "
              f"{tf:6.1f} TF  {tf / PEAK:5.1%}")
    print(f"TOTAL {tot_us:.1f} us, {tot_fl / tot_us / 1e6:.1f} TF ({tot_fl / tot_us / 1e6 / PEAK:.1%})")


if __name__ == "__main__":
    main()
