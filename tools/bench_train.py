#!/usr/bin/env python3
"""Secondary metric: train frames/s = B*(n_past+n_future-1)/iteration for
train_model + finetune_temporal_encoders (train.py:354-361), plus a per-kernel time breakdown."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import train  # noqa: E402
import utils  # noqa: E402
from dvg_amd import ops  # noqa: E402
from dvg_amd.data import SyntheticMovingMNIST  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dcgan")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n_past", type=int, default=10)
    ap.add_argument("--n_future", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--no_ft", action="store_true")
    a = ap.parse_args()
    o = train.build_parser().parse_args(["--model", a.model, "--batch_size", str(a.batch), "--n_past", str(a.n_past),
                                         "--n_future", str(a.n_future), "--no_save"])
    o.ft, o.rank, o.world, o.local_batch = not a.no_ft, 0, 1, a.batch
    torch.manual_seed(1)
    tr = train.Trainer(o, torch.device("cuda:0"))
    tr.train_mode()
    seq = SyntheticMovingMNIST(seq_len=a.n_past + a.n_future, seed=1).batch(a.batch)
    x, _ = utils.normalize_data(o, torch.cuda.FloatTensor, seq)

    def it():
        tr.train_model(x)
        if o.ft:
            tr.finetune_temporal_encoders(x)
    it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    it()
    ops.set_timer(None)
    agg = timer.summary()
    tot = sum(v["ms"] for v in agg.values())
    out = {"model": a.model, "batch": a.batch, "T": a.n_past + a.n_future, "ms_per_iter": round(dt * 1e3, 1),
           "train_frames_per_s": round(a.batch * (a.n_past + a.n_future - 1) / dt, 1),
           "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1), "timed_kernel_ms": round(tot, 1),
           "kernels": {k: {"n": v["launches"], "ms": round(v["ms"], 1),
                           "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)}
                       for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
