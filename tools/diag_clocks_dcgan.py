#!/usr/bin/env python3
"""Clock stamps (dvg_debug_set_clockbuf) for the dcgan_64 igemm layers: conv4x4s2 and convT4x4s2 at B=64."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from dvg_amd._lib import LIB_PATH  # noqa: E402

LAYERS = [("c4s2", 32, 64, 0, 128), ("c4s2", 16, 128, 0, 256), ("c4s2", 8, 256, 0, 512),
          ("cT", 4, 512, 512, 256), ("cT", 8, 256, 256, 128), ("cT", 16, 128, 128, 64),
          # x halves of the vgg_64 decoder's upsample convs (as transposed convs, fused._upconv_packed)
          ("cT", 4, 512, 0, 512), ("cT", 8, 256, 0, 256), ("cT", 16, 128, 0, 128), ("cT", 32, 64, 0, 64)]
WARM_S = float(os.environ.get("DIAG_WARM_S", "0.3"))


def main():
    lib = ctypes.CDLL(LIB_PATH)
    lib.dvg_debug_set_clockbuf.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    dev = torch.device("cuda:0")
    N = int(os.environ.get("DIAG_BATCH", "64"))
    buf = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
    for kind, H, C1, C2, Cout in LAYERS:
        x = ops.nhwc_empty(N, C1, H, H, dev).normal_()
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        if kind == "c4s2":
            wp = ops.pack_igemm_weight(torch.randn(Cout, C1, 4, 4, device=dev) * 0.02)
            fn = lambda: ops.conv4x4s2(x, wp, sc, sh)
            fl = 2.0 * N * (H // 2) ** 2 * Cout * 16 * C1
            stages = (C1 // 16) * 2
        else:
            sk = ops.nhwc_empty(N, C2, H, H, dev).normal_() if C2 else None
            wp = ops.pack_igemm_weight(torch.randn(C1 + C2, Cout, 4, 4, device=dev) * 0.02, transposed=True)
            fn = lambda: ops.convT4x4s2(x, sk, wp, sc, sh)
            fl = 2.0 * N * H * H * Cout * 16 * (C1 + C2)
            stages = (C1 + C2) // 16
        t0 = time.time()
        while time.time() - t0 < WARM_S:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        buf.zero_()
        lib.dvg_debug_set_clockbuf(ctypes.c_void_p(buf.data_ptr()), buf.numel() // 8)
        fn()
        torch.cuda.synchronize()
        lib.dvg_debug_set_clockbuf(ctypes.c_void_p(0), 0)
        d = buf.cpu().numpy().reshape(-1, 8)
        d = d[d[:, 3] != 0]
        c0, c1, c2, c3, w0, w1 = [d[:, i].astype(np.float64) for i in range(6)]
        span = (w1.max() - w0.min()) / 100.0
        print(f"{kind} {H}x{H} Cin {C1 + C2} Cout {Cout}: {len(d)} wgs, {us:.1f} us/launch = {fl / us / 1e6:.1f} TF "
              f"({fl / us / 1e6 / 157.3:.1%}); wg span {span:.1f} us")
        mhz = ((c3 - c0) / np.maximum(w1 - w0, 1) * 100.0)
        print(f"   shader clock during the kernel: {np.median(mhz):.0f} MHz (clock64 cycles / wall_clock64 ticks @100 MHz)")
        st, en = (w0 - w0.min()) / 100.0, (w1 - w0.min()) / 100.0
        pc = lambda a: " ".join(f"{np.percentile(a, q):5.1f}" for q in (0, 10, 50, 90, 100))   # noqa: E731
        print(f"   wg start us (p0 p10 p50 p90 p100): {pc(st)}   end: {pc(en)}   duration: {pc(en - st)}")
        if os.environ.get("DIAG_PAIRS"):
            # co-residency: group workgroups by (XCC, SE, CU) from HW_ID (bits 8-11 CU, 13-15 SE on gfx9) and compare the
            # durations inside each CU
            hw = d[:, 7].astype(np.int64)
            key = d[:, 6].astype(np.int64) * 4096 + ((hw >> 8) & 0xF) + 16 * ((hw >> 13) & 0x7) + 256 * ((hw >> 12) & 1)
            dur = (w1 - w0) / 100.0
            groups = {}
            for k, t, s0 in zip(key, dur, w0):
                groups.setdefault(int(k), []).append((s0, t))
            sizes = np.bincount([len(v) for v in groups.values()])
            firsts = [sorted(v)[0][1] for v in groups.values() if len(v) == 2]
            seconds = [sorted(v)[1][1] for v in groups.values() if len(v) == 2]
            print(f"   CUs by resident workgroups {dict(enumerate(sizes))}; pairs: first-started {np.mean(firsts):.1f} us, "
                  f"second-started {np.mean(seconds):.1f} us, |diff| mean {np.mean(np.abs(np.array(firsts) - np.array(seconds))):.1f} us")
        nst = (c2 - c1).mean() / max(1, stages)
        print(f"   cycles: prologue {np.mean(c1 - c0):7.0f}  loop {np.mean(c2 - c1):8.0f}  epilogue {np.mean(c3 - c2):7.0f}"
              f"   (loop / (chunks x stages-per-chunk) = {nst:.0f})")


if __name__ == "__main__":
    main()
