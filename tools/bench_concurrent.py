#!/usr/bin/env python3
"""Complete rollouts (condition + sample, B = 64 each) replayed CONCURRENTLY: n hipGraphs, each with its own static buffers
and private pool, on n streams.  The rollouts of make_gifs' `for s in range(nsample)` loop (generate_frames.py:143-177) are
independent, so the latency-bound phases of one (LSTM warm-up, GP sample, launch tails of the small layers) can fill with the
MFMA-bound phases of another.  Prints frames/s for n = 1, 2, 3 in flight (GPU only)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dvg_amd.data import SyntheticMovingMNIST  # noqa: E402
from dvg_amd.rollout import ConcurrentRollouts  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vgg")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--inflight", default="1,2,3")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    enc, dec, fp, gp, lik = bench.build_models(a.model, 64, 1, dev, 1)
    x = SyntheticMovingMNIST(seq_len=20, seed=1).batch_device(64, dev)
    bench.calibrate_batchnorm(enc, dec, x[0])
    for n in [int(s) for s in a.inflight.split(",")]:
        cr = ConcurrentRollouts(enc, dec, fp, gp, lik, x, 10, 20, inflight=n)
        for _ in range(2):
            cr.run(2 * n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cr.run(a.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"model": a.model, "inflight": n, "ms_per_rollout": round(1e3 * dt / a.steps, 3),
                          "frames_per_s": round(64 * 10 * a.steps / dt, 1)}), flush=True)
        del cr


if __name__ == "__main__":
    main()
