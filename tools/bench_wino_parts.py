#!/usr/bin/env python3
"""Per-kernel time of every Winograd-form layer of a vgg_64 rollout step (B = 64) and of the conditioning batch (B = 576):
input transform, batched GEMM, output transform, fused output->input transform - each timed alone, back to back, at steady
clocks (GPU only).  Columns: us, and the rate against the roof that binds the kernel (TFLOP/s executed, TB/s algorithmic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from dvg_amd._lib import lib  # noqa: E402
from tools.bench_small import time_fn  # noqa: E402

# (name, H, Cin, Cout, pool, chained_out)
LAYERS = [("c2.0", 32, 64, 128, False, True), ("c2.1", 32, 128, 128, True, False),
          ("c3.0", 16, 128, 256, False, True), ("c3.1", 16, 256, 256, False, True), ("c3.2", 16, 256, 256, True, False),
          ("c4.0", 8, 256, 512, False, True), ("c4.1", 8, 512, 512, False, True), ("c4.2", 8, 512, 512, True, False),
          ("upc2.1", 8, 512, 512, False, True), ("upc2.2", 8, 512, 256, False, False),
          ("upc3.1", 16, 256, 256, False, True), ("upc3.2", 16, 256, 128, False, False),
          ("upc4.1", 32, 128, 64, False, False)]


def main():
    dev = torch.device("cuda:0")
    p = ops._p
    batches = [int(b) for b in os.environ.get("BENCH_BATCHES", "64,576").split(",")]
    for N in batches:
        tot = {"in": 0.0, "gemm": 0.0, "out": 0.0, "fused": 0.0}
        print(f"--- B = {N}")
        for name, H, C, Cout, pool, chained in LAYERS:
            if N > 64 and name.startswith("upc"):
                continue
            if not ops.winograd_ok(N, C, H, H, Cout, 4):
                print(f"{name}: not a Winograd shape")
                continue
            T = N * (H // 4) ** 2
            x = ops.nhwc_empty(N, C, H, H, dev).normal_()
            # BENCH_V: what the GEMM's A operand holds - "input" (default: the real input transform of random data, written
            # by the timed dvg_winograd_input call below), "zeros" or "randn" (DVFS: an MFMA loop on zeros holds a higher clock)
            v = torch.empty((36, T, C), device=dev)
            m = torch.empty((36, T, Cout), device=dev)
            u = ops.winograd_weight(torch.randn(Cout, C, 3, 3, device=dev) * 0.02, 4)
            sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
            y = ops.nhwc_empty(N, Cout, H, H, dev)
            yp = ops.nhwc_empty(N, Cout, H // 2, H // 2, dev) if pool else None
            s = ops._stream
            t_in = time_fn(lambda: lib().dvg_winograd_input(p(x), p(v), N, H, H, C, 4, 0, s()))
            mode_v = os.environ.get("BENCH_V", "input")
            if mode_v == "zeros":
                v.zero_()
                u.zero_()
            elif mode_v == "randn":
                v.normal_()
            t_g = time_fn(lambda: lib().dvg_gemm_batched_k16(p(v), p(u), p(m), 36, T // 16, 16, C, Cout, s()))
            t_out = time_fn(lambda: lib().dvg_winograd_output(p(m), p(sc), p(sh), p(y), p(yp), N, H, H, Cout, 1, 0.2, 4, None, 0, s()))
            line = (f"{name:7s} {H:2d}x{H:<2d} {C:3d}->{Cout:3d}  in {t_in:7.1f} us {4e-6 * (x.numel() + v.numel()) / t_in:5.2f} TB/s | "
                    f"gemm {t_g:7.1f} us {2e-6 * 36 * T * C * Cout / t_g:6.1f} TF ({4e-6 * (v.numel() + m.numel() + u.numel()) / t_g:5.2f} TB/s) | "
                    f"out {t_out:7.1f} us {4e-6 * (m.numel() + y.numel() * (1.25 if pool else 1)) / t_out:5.2f} TB/s")
            tot["gemm"] += t_g
            if ops.winograd_chain_ok(N, Cout, H, H):
                vn = torch.empty((36, T, Cout), device=dev)
                t_f = time_fn(lambda: lib().dvg_winograd_output_input(p(m), p(sc), p(sh), p(vn), N, H, H, Cout, 1, 0.2, None, s()))
                line += f" | fused {t_f:7.1f} us {4e-6 * (m.numel() + vn.numel()) / t_f:5.2f} TB/s"
            else:
                t_f = None
            if pool and ops.winograd_pool_chain_ok(N, Cout, H, H):
                vp = torch.empty((36, T // 4, Cout), device=dev)
                t_p = time_fn(lambda: lib().dvg_winograd_output_pool_input(p(m), p(sc), p(sh), p(y), p(vp), N, H, H, Cout, 1, 0.2, 0, s()))
                line += f" | out+pool+in {t_p:7.1f} us {4e-6 * (m.numel() + y.numel() + vp.numel()) / t_p:5.2f} TB/s"
            # what the rollout runs today for this layer
            first = name.endswith(".0") or name.endswith("c2.1") or name in ("upc2.1", "upc3.1", "upc4.1")
            if name in ("c2.0", "c3.0", "c4.0", "upc2.1", "upc3.1", "upc4.1"):
                tot["in"] += t_in
            if chained and t_f is not None:
                tot["fused"] += t_f
            elif chained:
                tot["out"] += t_out
                tot["in"] += t_in      # the next layer's own input transform (same size class: c2.1 reads 128ch)
            else:
                tot["out"] += t_out
            print(line)
        print(f"    per pass: gemm {tot['gemm']:.0f} us, transforms {tot['in'] + tot['out'] + tot['fused']:.0f} us "
              f"(in {tot['in']:.0f}, out {tot['out']:.0f}, fused {tot['fused']:.0f})")


if __name__ == "__main__":
    main()
