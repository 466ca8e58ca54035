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
    ap.add_argument("--graph", action="store_true", help="train.GraphedIteration: the iteration as one hipGraph")
    ap.add_argument("--segmented", action="store_true",
                    help="train.SegmentedIteration: a chain of hipGraphs cut at the gradient all-reduces; with "
                         "DVG_FORCE_ALLREDUCE=1 a 1-rank RCCL group is created so that the collectives really run")
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--image_width", type=int, default=64)
    a = ap.parse_args()
    o = train.build_parser().parse_args(["--model", a.model, "--batch_size", str(a.batch), "--n_past", str(a.n_past),
                                         "--n_future", str(a.n_future), "--no_save", "--channels", str(a.channels),
                                         "--image_width", str(a.image_width)])
    o.ft, o.rank, o.world, o.local_batch = not a.no_ft, 0, 1, a.batch
    if os.environ.get("DVG_FORCE_ALLREDUCE") == "1":
        import torch.distributed as dist
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda:0"))
    torch.manual_seed(1)
    tr = train.Trainer(o, torch.device("cuda:0"))
    tr.train_mode()
    if a.channels == 1 and a.image_width == 64:
        seq = SyntheticMovingMNIST(seq_len=a.n_past + a.n_future, seed=1).batch(a.batch)
    else:   # BAIR / UCF-shaped synthetic clips
        from dvg_amd.data import synthetic_video
        seq = synthetic_video(a.batch, a.n_past + a.n_future, a.channels, a.image_width, seed=1)
    x, _ = utils.normalize_data(o, torch.cuda.FloatTensor, seq)

    def it():
        tr.train_model(x)
        if o.ft:
            tr.finetune_temporal_encoders(x)
    if a.graph or a.segmented:
        g = (train.SegmentedIteration if a.segmented else train.GraphedIteration)(tr, warmup=2)
        run = lambda: g(x)   # noqa: E731
        for _ in range(3):   # 2 eager warm-up iterations + the capture
            run()
    else:
        run = it
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    it()
    ops.set_timer(None)
    agg = timer.summary()
    tot = sum(v["ms"] for v in agg.values())
    out = {"launch": ("hipGraph segments (%d) + eager all-reduces" % g.n_segments) if a.segmented else
           "hipGraph replay" if a.graph else "eager", "model": f"{a.model}_{a.image_width}", "channels": a.channels, "batch": a.batch, "T": a.n_past + a.n_future, "ms_per_iter": round(dt * 1e3, 1),
           "train_frames_per_s": round(a.batch * (a.n_past + a.n_future - 1) / dt, 1),
           "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1), "timed_kernel_ms": round(tot, 1),
           "kernels": {k: {"n": v["launches"], "ms": round(v["ms"], 1),
                           "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)}
                       for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
