#!/usr/bin/env python3
"""Per-launch cost (kernel + inter-kernel gap) of the small kernels under hipGraph replay: a graph of 100 dependent
launches of each, replayed at steady clocks (GPU only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402


def graph_time(fn, n=100, reps=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    t0 = time.time()
    while time.time() - t0 < 0.3:
        g.replay()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B = 64
    r = lambda *s: torch.randn(*s, device=dev)
    x, h, c = r(B, 256), r(B, 256), r(B, 256)
    wi, wh, bi, bh = r(1024, 256), r(1024, 256), r(1024), r(1024)
    state = {"h": h, "c": c}

    def cell():
        state["h"], state["c"] = ops.lstm_cell(x, state["h"], state["c"], wi, wh, bi, bh)
    print(f"lstm_cell (chain)            {graph_time(cell):7.2f} us/launch")
    a90, w_e = r(B, 90), r(256, 90)
    print(f"gemm embed (64,256,90)       {graph_time(lambda: ops.gemm_nt(a90, w_e, None, None)):7.2f} us/launch")
    a256, w_o = r(B, 256), r(90, 256)
    print(f"gemm output (64,90,256)      {graph_time(lambda: ops.gemm_nt(a256, w_o, None, None)):7.2f} us/launch")
    a8k, w8k = r(B, 8192), r(90, 8192)
    print(f"gemm c5 (64,90,8192) sk16    {graph_time(lambda: ops.gemm_nt(a8k, w8k, None, None, splitk=16)):7.2f} us/launch (2 kernels)")
    w_up = r(8192, 90)
    print(f"gemm upc1 (64,8192,90)       {graph_time(lambda: ops.gemm_nt(a90, w_up, None, None)):7.2f} us/launch")
    big, w9 = r(B * 4096, 64), r(9, 64)
    print(f"gemm last proj (262144,9,64) {graph_time(lambda: ops.gemm_nt(big, w9, None, None), n=20):7.2f} us/launch")
    xf, wf = torch.rand(B, 1, 64, 64, device=dev), r(64, 1, 3, 3) * 0.1
    sc, sh = torch.rand(64, device=dev) + 0.5, r(64) * 0.1
    us = graph_time(lambda: ops.conv3x3_first(xf, wf, sc, sh), n=20)
    print(f"conv3x3_first (64,1,64,64)->64 {us:7.2f} us/launch  {(xf.numel() + B * 4096 * 64) * 4 / us / 1e3:7.1f} GB/s")
    xp = ops.nhwc_empty(B, 64, 64, 64, dev).normal_()
    w9 = r(9, 64)
    us = graph_time(lambda: ops.pixel_proj(xp, w9), n=20)
    print(f"pixel_proj (262144,9,64)     {us:7.2f} us/launch  {(xp.numel() + B * 4096 * 9) * 4 / us / 1e3:7.1f} GB/s")
    wx, bias = r(1024, 92), r(1024)
    x90 = r(B, 90)

    def cellx():
        state["h"], state["c"] = ops.lstm_cell_x(x90, state["h"], state["c"], wx, wh, bias)
    print(f"lstm_cell_x (chain)          {graph_time(cellx):7.2f} us/launch")
    wt = torch.zeros(96, 8192, device=dev)
    wt[:90].normal_()
    out = torch.empty(B, 8192, device=dev)
    sc8, sh8 = torch.rand(512, device=dev) + 0.5, r(512) * 0.1
    print(f"stem_gemm (64,8192,90)       {graph_time(lambda: ops.stem_gemm(a90, wt, 90, sc8, sh8, out, period=512)):7.2f} us/launch")
    x4 = ops.nhwc_empty(B, 512, 4, 4, dev).normal_()
    from dvg_amd import fused
    import torch.nn as nn
    conv, bn = nn.Conv2d(512, 90, 4, 1, 0).to(dev), nn.BatchNorm2d(90).to(dev).eval()
    with torch.no_grad():
        print(f"encoder head (fused.head_bn_tanh) {graph_time(lambda: fused.head_bn_tanh(conv, bn, x4)):7.2f} us/call (2 kernels)")
    x1 = torch.rand(B, 1, 64, 64, device=dev)
    w4 = r(64, 1, 4, 4) * 0.1
    us = graph_time(lambda: ops.conv4x4s2_first(x1, w4, sc, sh), n=20)
    print(f"conv4x4s2_first (64,1,64,64)->64 {us:7.2f} us/launch  {(x1.numel() + B * 1024 * 64) * 4 / us / 1e3:7.1f} GB/s")
    t = torch.zeros(64, device=dev)
    print(f"torch fill (64 floats)       {graph_time(lambda: t.fill_(1.0)):7.2f} us/launch")


if __name__ == "__main__":
    main()
