#!/usr/bin/env python3
"""Steady-clock timing of the small (non-igemm) launches of one B=64 rollout step, per shape (GPU only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402


def time_fn(fn, iters=200, warm_s=0.3):
    t0 = time.time()
    while time.time() - t0 < warm_s:
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    B = 64
    r = lambda *s: torch.randn(*s, device=dev)
    rows = []
    for name, (m, n, k, sk) in {
        "lstm embed  (64,256,90)": (B, 256, 90, 1), "lstm output (64,90,256)": (B, 90, 256, 1),
        "enc head c5 (64,90,8192) splitk16": (B, 90, 8192, 16), "enc head c5 (640,90,8192) splitk16": (640, 90, 8192, 16),
        "dec stem upc1 (64,8192,90)": (B, 8192, 90, 1),
        "last proj (262144,9,64)": (B * 4096, 9, 64, 1), "dcgan last proj (65536,16,64)": (B * 1024, 16, 64, 1),
    }.items():
        a, w = r(m, k), r(n, k)
        rows.append((name, time_fn(lambda: ops.gemm_nt(a, w, None, None, splitk=sk)), 4.0 * (m * k + n * k + m * n)))
    x, h, c = r(B, 256), r(B, 256), r(B, 256)
    wi, wh, bi, bh = r(1024, 256), r(1024, 256), r(1024), r(1024)
    rows.append(("lstm_cell (64,256)", time_fn(lambda: ops.lstm_cell(x, h, c, wi, wh, bi, bh)), 4.0 * 2 * 1024 * 256))
    for name, us, byts in rows:
        print(f"{name:40s} {us:8.1f} us   {byts / us / 1e3:8.1f} GB/s")


if __name__ == "__main__":
    main()
