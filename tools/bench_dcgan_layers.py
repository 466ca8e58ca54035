#!/usr/bin/env python3
"""Per-layer timing of dcgan_64's stride-2 convs (dcgan_64.py:28-47: c2..c4) and transposed convs (upc2..upc4 with the skip
concat) in isolation, at the rollout step's batch and at a training pass's (GPU only): us, executed TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from tools.bench_layers import time_fn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="64,1280")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for n in [int(v) for v in a.batches.split(",")]:
        tot = 0.0
        for (h, cin, cout) in ((32, 64, 128), (16, 128, 256), (8, 256, 512)):
            x = ops.nhwc_empty(n, cin, h, h, dev).normal_()
            wp = ops.pack_igemm_weight(torch.randn(cout, cin, 4, 4, device=dev) * 0.02)
            sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
            us = time_fn(lambda: ops.conv4x4s2(x, wp, sc, sh))
            fl = 2.0 * n * (h // 2) ** 2 * cout * 16 * cin
            tot += us
            print(f"N={n:5d} conv4x4s2  {h:2d}x{h:<2d} {cin:3d}->{cout:3d}: {us:8.1f} us {fl / us / 1e6:7.1f} TF  split-K {ops.lib().dvg_conv_splitk_v2(1, n, h, h, cin, cout)}")
        for (h, c1, c2, cout) in ((4, 512, 512, 256), (8, 256, 256, 128), (16, 128, 128, 64)):
            x, sk = ops.nhwc_empty(n, c1, h, h, dev).normal_(), ops.nhwc_empty(n, c2, h, h, dev).normal_()
            wp = ops.pack_igemm_weight(torch.randn(c1 + c2, cout, 4, 4, device=dev) * 0.02, transposed=True)
            sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
            us = time_fn(lambda: ops.convT4x4s2(x, sk, wp, sc, sh))
            fl = 2.0 * n * (2 * h) ** 2 * cout * 4 * (c1 + c2)
            tot += us
            print(f"N={n:5d} convT4x4s2 {h:2d}x{h:<2d} {c1 + c2:4d}->{cout:3d}: {us:8.1f} us {fl / us / 1e6:7.1f} TF")
        print(f"N={n:5d} sum {tot:8.1f} us")


if __name__ == "__main__":
    main()
