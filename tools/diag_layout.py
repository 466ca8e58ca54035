#!/usr/bin/env python3
"""Which call sites trigger layout conversions / contiguous copies in one training iteration (GPU only)."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops  # noqa: E402

hits = collections.Counter()
orig_nhwc, orig_nchw = ops.to_nhwc, ops.to_nchw


def site():
    st = traceback.extract_stack()[:-2]
    keep = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in st if "/repo/" in f.filename and "diag_layout" not in f.filename]
    return " < ".join(reversed(keep[-4:]))


def to_nhwc(t):
    if not ops.is_nhwc(t):
        hits[("to_nhwc", tuple(t.shape), site())] += 1
    return orig_nhwc(t)


def to_nchw(t):
    if not t.is_contiguous():
        hits[("to_nchw", tuple(t.shape), site())] += 1
    return orig_nchw(t)


def main():
    import train
    import utils
    from dvg_amd.data import SyntheticMovingMNIST
    model = sys.argv[1] if len(sys.argv) > 1 else "dcgan"
    o = train.build_parser().parse_args(["--model", model, "--batch_size", "64", "--n_past", "10", "--n_future", "10", "--no_save"])
    o.ft, o.rank, o.world, o.local_batch = True, 0, 1, 64
    torch.manual_seed(1)
    tr = train.Trainer(o, torch.device("cuda:0"))
    tr.train_mode()
    seq = SyntheticMovingMNIST(seq_len=20, seed=1).batch(64)
    x, _ = utils.normalize_data(o, torch.cuda.FloatTensor, seq)
    tr.iteration(x)
    torch.cuda.synchronize()
    ops.to_nhwc, ops.to_nchw = to_nhwc, to_nchw
    tr.iteration(x)
    torch.cuda.synchronize()
    print("conversions in one iteration:", sum(hits.values()))
    for (kind, shape, where), n in hits.most_common(40):
        print(f"{n:5d} {kind} {shape} {where}")


if __name__ == "__main__":
    main()
