#!/usr/bin/env python3
"""Which Python lines of the training iteration launch torch (at::native / rocclr) kernels: one eager iteration under a
TorchDispatchMode that records, for every aten op that is not a pure view, the innermost Python frame inside this repository
(autograd's worker thread inherits the mode: backward nodes show up under the Function.backward that called them, built-in
nodes - AccumulateGrad, the backward of torch ops the closures compose - under "<autograd engine>").  GPU only.
usage: trace_torch_ops.py [--model dcgan] [--batch 16] [--n_past 2] [--n_future 10] [--channels 3]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VIEWS = {"view", "_unsafe_view", "reshape", "as_strided", "t", "transpose", "permute", "slice", "select", "detach", "alias",
         "expand", "unsqueeze", "squeeze", "split", "split_with_sizes", "unbind", "narrow", "view_as", "_reshape_alias",
         "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "set_", "is_same_size", "size", "stride",
         "sym_size", "unsafe_split", "chunk", "lift_fresh", "_local_scalar_dense", "item", "is_nonzero", "resize_",
         "record_stream", "_to_copy" * 0 + "numel", "dim", "is_pinned", "storage_offset", "is_contiguous", "sym_stride",
         "sym_numel", "sym_storage_offset", "unsafe_chunk", "squeeze_", "unsqueeze_", "t_", "transpose_", "result_type"}


class Tracer(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()
        self.elems = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in VIEWS:
            return out
        ts = [t for t in (list(args) + [out] + (list(out) if isinstance(out, (tuple, list)) else []))
              if torch.is_tensor(t) and t.is_cuda]
        if not ts:
            return out
        site = "<autograd engine>"
        for fr in reversed(traceback.extract_stack()[:-1]):
            f = fr.filename
            if ("dvg_amd/" in f or f.endswith("train.py") or f.endswith("gp_models.py")) and "dist-packages" not in f:
                site = f"{f.split('/repo/')[-1] if '/repo/' in f else os.path.basename(f)}:{fr.lineno} {fr.name}"
                break
        self.count[(site, name)] += 1
        self.elems[(site, name)] += max(t.numel() for t in ts)
        return out


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
    tracer = Tracer()
    with tracer:
        tr.iteration(x)
        torch.cuda.synchronize()
    print(f"{sum(tracer.count.values())} non-view aten ops on device tensors in one iteration")
    for (site, name), n in tracer.count.most_common(80):
        print(f"{n:5d}  {tracer.elems[(site, name)] / n:12.0f} elems/op  {name:24s} {site}")


if __name__ == "__main__":
    main()
