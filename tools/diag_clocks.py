#!/usr/bin/env python3
"""Diagnostic: per-workgroup clock stamps of the v2 igemm kernel (debug hook dvg_debug_set_clockbuf).
Prints the shader clock during the kernel (clock64 vs the 100 MHz wall_clock64), the distribution of
workgroup phases (prologue / main loop / epilogue, in shader cycles) and the launch span."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from dvg_amd._lib import LIB_PATH  # noqa: E402

LAYERS = [(64, 64, 0, 64, 0, 1), (16, 256, 0, 256, 0, 0), (8, 512, 0, 512, 0, 0), (8, 512, 512, 512, 1, 0),
          (32, 128, 128, 128, 1, 0)]


WARM_S = float(os.environ.get('DIAG_WARM_S', '2.0'))


def main():
    lib = ctypes.CDLL(LIB_PATH)
    lib.dvg_debug_set_clockbuf.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    dev = torch.device("cuda:0")
    N = 64
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    for (H, C1, C2, Cout, up, pool) in LAYERS:
        hx = H // 2 if up else H
        x = ops.nhwc_empty(N, C1, hx, hx, dev).normal_()
        sk = ops.nhwc_empty(N, C2, H, H, dev).normal_() if C2 else None
        wp = ops.pack_igemm_weight(torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.02)
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        fn = lambda: ops.conv3x3(x, sk, wp, sc, sh, upsample=bool(up), pool=bool(pool))
        import time
        t0 = time.time()
        while time.time() - t0 < WARM_S:   # sustained load so that the clock reaches its steady state
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
        buf.zero_()
        lib.dvg_debug_set_clockbuf(ctypes.c_void_p(buf.data_ptr()), buf.numel() // 8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        lib.dvg_debug_set_clockbuf(ctypes.c_void_p(0), 0)
        d = buf.cpu().numpy().reshape(-1, 8)
        d = d[d[:, 3] != 0]
        nwg = len(d)
        c0, c1, c2, c3, w0, w1 = [d[:, i].astype(np.float64) for i in range(6)]
        ghz = ((c3 - c0) / ((w1 - w0) / 100e6)).mean() / 1e9
        span_us = (w1.max() - w0.min()) / 100.0
        fl = 2.0 * N * H * H * Cout * 9 * (C1 + C2)
        ksteps = (C1 + C2) // 16
        print(f"conv3x3 {H}x{H} Cin {C1 + C2} Cout {Cout}: {nwg} wgs, event {e0.elapsed_time(e1) * 1e3:.1f} us, "
              f"wg span {span_us:.1f} us, shader clock {ghz:.3f} GHz -> peak at this clock "
              f"{157.3 * ghz / 2.4:.1f} TF, achieved {fl / span_us / 1e6:.1f} TF")
        print(f"   cycles: prologue {np.mean(c1 - c0):8.0f}  loop {np.mean(c2 - c1):8.0f} "
              f"({np.mean(c2 - c1) / ksteps:.0f}/stage, min {np.min(c2 - c1) / ksteps:.0f} max {np.max(c2 - c1) / ksteps:.0f})"
              f"  epilogue {np.mean(c3 - c2):8.0f}   start skew {(w0.max() - w0.min()) / 100.0:.1f} us")
        xcc = d[:, 6] & 0xf
        print("   wgs per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())
        if os.environ.get("DIAG_HWID"):
            full = buf.cpu().numpy().reshape(-1, 8)
            hw = full[:, 7]
            wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
            from collections import defaultdict
            g = defaultdict(list)
            for b in range(nwg):
                if full[b, 3]:
                    g[(int(full[b, 6] & 15), int(se[b]), int(sh[b]), int(cu[b]))].append((b, int(wave[b]), int(simd[b]), int(full[b, 4] - w0.min())))
            for k in sorted(g)[:6]:
                print("   CU", k, g[k])
            print("   distinct CUs:", len(g))


if __name__ == "__main__":
    main()
