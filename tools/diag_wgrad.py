#!/usr/bin/env python3
"""Phase stamps of wgrad_igemm_kernel (debug hook dvg_debug_set_wgrad_clockbuf): per-workgroup cycles spent staging
tiles vs in the MFMA loop vs writing the partials, for the vgg_64 layer shapes at B=64."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops
from dvg_amd._lib import LIB_PATH
lib = ctypes.CDLL(LIB_PATH); lib.dvg_debug_set_wgrad_clockbuf.argtypes = [ctypes.c_void_p, ctypes.c_uint]
dev = torch.device("cuda:0"); N = 64
for (H, Cin, Cout) in [(64, 64, 64), (32, 128, 128), (16, 256, 256), (8, 512, 512)]:
    x = ops.nhwc_empty(N, Cin, H, H, dev).normal_(); du = ops.nhwc_empty(N, Cout, H, H, dev).normal_()
    fn = lambda: ops.conv_wgrad(ops.MODE_CONV3, x, None, du)
    t0 = time.time()
    while time.time() - t0 < 0.4:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    buf = torch.zeros(4096 * 4, dtype=torch.int64, device=dev)
    lib.dvg_debug_set_wgrad_clockbuf(ctypes.c_void_p(buf.data_ptr()), buf.numel() // 4); fn(); torch.cuda.synchronize(); lib.dvg_debug_set_wgrad_clockbuf(ctypes.c_void_p(0), 0)
    d = buf.cpu().numpy().reshape(-1, 4).astype(np.float64); d = d[d[:, 3] > 0]
    fl = 2.0 * N * H * H * Cout * 9 * Cin
    tiles = d[:, 3].mean()
    print(f"wgrad {H}x{H} {Cin}->{Cout}: {us:.1f} us (incl. reduce) = {fl / us / 1e6:.1f} TF; {len(d)} wgs x {tiles:.1f} tiles; per tile: staging {d[:,0].mean()/tiles:.0f}  mfma {d[:,1].mean()/tiles:.0f} (ideal 36864 x2 wgs/CU)  final partial write {d[:,2].mean():.0f} cycles")
