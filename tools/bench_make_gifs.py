#!/usr/bin/env python3
"""make_gifs (generate_frames.py:143-189) throughput: nsample rollouts of one batch + SSIM / PSNR + best-of-N, eager sample
loop vs the replayed sample body with 1 / 3 samples in flight (GPU only)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generate_frames  # noqa: E402
from dvg_amd.data import SyntheticMovingMNIST  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dcgan")
    ap.add_argument("--nsample", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for inflight in (0, 1, 3):
        opt = generate_frames.build_parser().parse_args(["--synthetic_ckpt", "--batch_size", "64", "--model", a.model,
                                                         "--n_past", "10", "--n_eval", "20", "--inflight", str(inflight)])
        torch.manual_seed(1)
        g = generate_frames.Generator(opt, generate_frames.synthetic_checkpoint(opt), dev)
        x = SyntheticMovingMNIST(seq_len=20, seed=1).batch_device(64, dev)
        g.make_gifs(x, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.make_gifs(x, a.nsample)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"model": a.model, "inflight": inflight, "nsample": a.nsample, "s_per_batch": round(dt, 4),
                          "predicted_frames_per_s": round(64 * 10 * a.nsample / dt, 1)}), flush=True)


if __name__ == "__main__":
    main()
