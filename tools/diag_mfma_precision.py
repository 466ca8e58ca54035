#!/usr/bin/env python3
"""Error of the 3x3 layers against an fp64 convolution, per form (direct implicit GEMM, Winograd F(4x4)): run it under
two builds of the library (DVG_HIP_LIB=...) to compare the native fp32 MFMA with the 3 x bf16 split on the bf16 pipe.
    python tools/diag_mfma_precision.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (N, H, C, Cout) in [(8, 64, 64, 64), (8, 32, 128, 128), (32, 16, 256, 256), (64, 8, 512, 512)]:
        xn = torch.randn(N, C, H, H, device=dev)
        x = ops.to_nhwc(xn)
        w = torch.randn(Cout, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
        ref = F.conv2d(xn.double(), w.double(), padding=1)
        one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        y = ops.conv3x3(x, None, ops.pack_igemm_weight(w), one, zero, act=ops.ACT_NONE)
        ed = (y.double() - ref).abs()
        line = f"N {N:2d} {H:2d}x{H:<2d} {C:3d}->{Cout:3d}: direct max {float(ed.max() / ref.abs().max()):.2e} rms {float(ed.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e} bias {float((y.double() - ref).mean() / ref.abs().mean()):+.1e}"
        if ops.winograd_ok(N, C, H, H, Cout, 4):
            yw = ops.conv3x3_winograd(x, ops.winograd_weight(w, 4), one, zero, act=ops.ACT_NONE)
            ew = (yw.double() - ref).abs()
            line += f" | F4 max {float(ew.max() / ref.abs().max()):.2e} rms {float(ew.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e} bias {float((yw.double() - ref).mean() / ref.abs().mean()):+.1e}"
        # torch's own fp32 conv as the yardstick
        yt = F.conv2d(xn, w, padding=1)
        et = (yt.double() - ref).abs()
        line += f" | torch fp32 max {float(et.max() / ref.abs().max()):.2e} rms {float(et.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
