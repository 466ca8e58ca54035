#!/usr/bin/env python3
"""Which Python lines of the training iteration launch torch (at::native / rocclr) kernels: one eager iteration under
torch.profiler with stacks, aten ops that launched a device kernel grouped by the innermost frame inside this repository.
GPU only.  usage: trace_torch_ops.py [--model dcgan] [--batch 16] [--n_past 2] [--n_future 10] [--channels 3]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dcgan")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--n_past", type=int, default=2)
    ap.add_argument("--n_future", type=int, default=10)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--image_width", type=int, default=64)
    a = ap.parse_args()
    import train
    import utils
    from dvg_amd.data import synthetic_video
    T = a.n_past + a.n_future
    opt = train.build_parser().parse_args(["--model", a.model, "--channels", str(a.channels), "--image_width", str(a.image_width),
                                           "--dataset", "bair", "--batch_size", str(a.batch), "--n_past", str(a.n_past),
                                           "--n_future", str(a.n_future), "--no_save", "--synthetic_data"])
    opt.ft, opt.rank, opt.world, opt.local_batch = True, 0, 1, a.batch
    torch.manual_seed(1)
    tr = train.Trainer(opt, torch.device("cuda:0"))
    tr.train_mode()
    x, _ = utils.normalize_data(opt, torch.cuda.FloatTensor, synthetic_video(a.batch, T, a.channels, a.image_width, seed=1))
    for _ in range(3):
        tr.iteration(x)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.iteration(x)
        torch.cuda.synchronize()
    by_site = collections.Counter()
    by_site_us = collections.Counter()
    n_kernels = 0
    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        # leaf aten ops only: an op whose child also launched the same kernels would be counted twice
        if any(c.name.startswith("aten::") and c.kernels for c in ev.cpu_children):
            continue
        site = "?"
        for fr in (ev.stack or []):
            if ("dvg_amd/" in fr or "train.py" in fr or "gp_models.py" in fr) and "site-packages" not in fr and "dist-packages" not in fr:
                site = fr.split("/repo/")[-1]
                break
        if site == "?" and ev.stack:
            site = "? " + " <- ".join(f.split("/")[-1] for f in ev.stack[:3])
        by_site[(site, ev.name)] += len(ev.kernels)
        by_site_us[(site, ev.name)] += sum(k.duration for k in ev.kernels)
        n_kernels += len(ev.kernels)
    print(f"{n_kernels} device kernels launched by aten ops in one iteration")
    for (site, name), n in by_site.most_common(60):
        print(f"{n:5d}  {by_site_us[(site, name)]:8.0f} us  {name:28s} {site}")


if __name__ == "__main__":
    main()
