#!/usr/bin/env python3
"""Clock stamps (dvg_debug_set_clockbuf) for the batched Winograd-domain GEMMs (igemm GEMM mode) at the vgg_64 shapes."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402
from dvg_amd._lib import LIB_PATH, lib  # noqa: E402

# (tiles per position, Cin, Cout): B = 64 rollout shapes, then the B = 576 conditioning pass
SHAPES = [(256, 512, 512), (1024, 256, 256), (4096, 128, 128), (1024, 256, 512), (4096, 128, 256),
          (2304, 512, 512), (9216, 256, 256), (36864, 128, 128)]


def main():
    dbg = ctypes.CDLL(LIB_PATH)
    dbg.dvg_debug_set_clockbuf.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    dev = torch.device("cuda:0")
    buf = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
    for t, c, co in SHAPES:
        v = torch.randn(36, t, c, device=dev)
        u = torch.randn(36, c // 16, 1, co, ops.packed_row_floats(), device=dev) * 0.05
        m = torch.empty(36, t, co, device=dev)
        fn = lambda: ops.check(lib().dvg_gemm_batched_k16(ops._p(v), ops._p(u), ops._p(m), 36, t // 16, 16, c, co,
                                                          ops._stream()), "gemm")   # noqa: E731
        fl = 2.0 * 36 * t * c * co
        t0 = time.time()
        while time.time() - t0 < 0.3:
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
        dbg.dvg_debug_set_clockbuf(ctypes.c_void_p(buf.data_ptr()), buf.numel() // 8)
        fn()
        torch.cuda.synchronize()
        dbg.dvg_debug_set_clockbuf(ctypes.c_void_p(0), 0)
        d = buf.cpu().numpy().reshape(-1, 8)
        d = d[d[:, 3] != 0]
        c0, c1, c2, c3, w0, w1 = [d[:, i].astype(np.float64) for i in range(6)]
        span = (w1.max() - w0.min()) / 100.0
        print(f"T {t} Cin {c} Cout {co}: {len(d)} wgs, {us:.1f} us/launch = {fl / us / 1e6:.1f} TF ({fl / us / 1e6 / 157.3:.1%});"
              f" wg span {span:.1f} us; ideal at peak {fl / 157.3e6:.1f} us")
        st, en = (w0 - w0.min()) / 100.0, (w1 - w0.min()) / 100.0
        pc = lambda a: " ".join(f"{np.percentile(a, q):5.1f}" for q in (0, 10, 50, 90, 100))   # noqa: E731
        print(f"   wg start us (p0 p10 p50 p90 p100): {pc(st)}   end: {pc(en)}   duration: {pc(en - st)}")
        hw = d[:, 7].astype(np.int64)
        key = d[:, 6].astype(np.int64) * 4096 + ((hw >> 8) & 0xF) + 16 * ((hw >> 13) & 0x7) + 256 * ((hw >> 12) & 1)
        groups = {}
        for k, a, b in zip(key, st, en):
            groups.setdefault(int(k), []).append((a, b))
        per_cu = np.array([len(g) for g in groups.values()])
        conc, busy_until = [], []
        for g in groups.values():
            ev = sorted([(a, 1) for a, _ in g] + [(b, -1) for _, b in g])
            cur = mx = 0
            for _, s in ev:
                cur += s
                mx = max(mx, cur)
            conc.append(mx)
            busy_until.append(max(b for _, b in g))
        print(f"   CUs used {len(groups)}; wgs per CU min/mean/max {per_cu.min()}/{per_cu.mean():.2f}/{per_cu.max()}; max concurrent "
              f"per CU {dict(zip(*np.unique(conc, return_counts=True)))}; CU last-end us (p0 p50 p100): "
              f"{np.percentile(busy_until, 0):.1f} {np.percentile(busy_until, 50):.1f} {np.percentile(busy_until, 100):.1f}")
        print(f"   cycles: prologue {np.mean(c1 - c0):7.0f}  loop {np.mean(c2 - c1):8.0f}  epilogue {np.mean(c3 - c2):7.0f};"
              f" shader clock {np.median((c3 - c0) / np.maximum(w1 - w0, 1) * 100.0):.0f} MHz")


if __name__ == "__main__":
    main()
