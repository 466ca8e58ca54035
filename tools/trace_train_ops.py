#!/usr/bin/env python3
"""Where do the small torch kernels of a training iteration come from?  One eager iteration (train_model + both fine-tuning
closures) at the C4 shape under torch.profiler with Python stacks; prints, per aten op that launches a kernel, the call sites
inside this repo (file:line) with their launch counts.  GPU only; diagnostic."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import train  # noqa: E402
import utils  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "dcgan"
    o = train.build_parser().parse_args(["--model", model, "--batch_size", "16", "--n_past", "2", "--n_future", "10", "--no_save",
                                         "--channels", "3", "--image_width", "64"])
    o.ft, o.rank, o.world, o.local_batch = True, 0, 1, 16
    torch.manual_seed(1)
    tr = train.Trainer(o, torch.device("cuda:0"))
    tr.train_mode()
    from dvg_amd.data import synthetic_video
    x, _ = utils.normalize_data(o, torch.cuda.FloatTensor, synthetic_video(16, 12, 3, 64, seed=1))

    def it():
        tr.train_model(x)
        tr.finetune_temporal_encoders(x)
    for _ in range(2):
        it()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        it()
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sites = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not getattr(ev, "kernels", None):
            continue
        site = "?"
        for fr in (ev.stack or []):
            if root in fr and "/tools/" not in fr:
                site = fr.replace(root + "/", "").split(",")[0]
                break
        sites[(ev.name, site)] += len(ev.kernels)
    for (name, site), n in sites.most_common(60):
        print(f"{n:4d}  {name:28s} {site}")


if __name__ == "__main__":
    main()
