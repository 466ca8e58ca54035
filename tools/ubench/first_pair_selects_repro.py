"""Reproducer for the r03 finding at dvg_conv3x3_first_pair: with the LeakyReLU / zero-padding of the
in-kernel first layer written as selects (-DDVG_FIRST_SELECTS=1: hipcc turns them into EXEC-masked blocks inside the
MFMA-interleaved stage loop) the launch gave run-to-run different tiles.  Run with DVG_HIP_LIB pointing at that build
(tools/ab_variants.sh conv_igemm2.hip DVG_FIRST_SELECTS 1) and at the shipped one; prints, per batch size, how many of
REPS launches differ from the first one, how many elements, and where (channel / pixel parity, image border or interior)."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvg_amd import fused  # noqa: E402

REPS = int(os.environ.get("REPS", "30"))
torch.manual_seed(0)
dev = torch.device("cuda:0")
conv0, bn0, conv1, bn1 = nn.Conv2d(1, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.Conv2d(64, 64, 3, 1, 1), nn.BatchNorm2d(64)
with torch.no_grad():
    for bn in (bn0, bn1):
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
for m in (conv0, bn0, conv1, bn1):
    m.to(dev).eval()
for N in (16, 50, 64):
    x = torch.rand(N, 1, 64, 64, device=dev)
    with torch.no_grad():
        h0 = fused.conv3_first_bn_act(conv0, bn0, x)
        ref, _ = fused.conv3_bn_act(conv1, bn1, h0, pool=True)          # the two-launch path
        first = fused.conv3_first_pair(conv0, bn0, conv1, bn1, x, pool=True)[0].clone()
        ndiff, worst, where = 0, 0, None
        for r in range(REPS):
            y = fused.conv3_first_pair(conv0, bn0, conv1, bn1, x, pool=True)[0]
            d = (y != first)
            if bool(d.any()):
                ndiff += 1
                cnt = int(d.sum())
                if cnt > worst:
                    worst = cnt
                    idx = d.nonzero()
                    where = {"images": sorted(set(idx[:, 0].tolist()))[:8], "channels_mod2": sorted(set((idx[:, 1] % 2).tolist())),
                             "rows": sorted(set(idx[:, 2].tolist()))[:12], "cols": sorted(set(idx[:, 3].tolist()))[:12],
                             "max_abs_diff": float((y - first).abs().max())}
        err = float((first - ref).abs().max() / ref.abs().max())
    print(f"lib={os.path.basename(os.environ.get('DVG_HIP_LIB', 'libdvg_hip.so'))} N={N}: {ndiff}/{REPS} launches differ from the first; "
          f"worst {worst} elements; rel err of the first launch vs the two-launch path {err:.2e}; {where}", flush=True)
