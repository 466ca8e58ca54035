#!/usr/bin/env python3
"""Which layers make the HIP path noisier than the reference's fp32 arithmetic?  (GPU only.)

At B = 50 (the reference's generate batch) the element-wise error (1 % floor) of the vgg_64 latent against the fp64 oracle was
1.9e-4 for the HIP path where the fp32 oracle - i.e. the reference's own arithmetic - has 8.3e-5, and the suspicion was the
rounding of the Winograd F(4x4,3x3) transforms.  This tool attributes it:

  1. END TO END, encoder and decoder: the error of the latent / the decoded frame against fp64 with (a) the default layer
     forms, (b) every 3x3 layer in direct form, (c) ONLY layer L taken out of F(4x4) (direct, and F(2x2) where that form
     exists) - how much of the excess each layer is responsible for - and (d) ONLY layer L in F(4x4), everything else direct;
  2. the fp32 oracle's own error beside every figure (the yardstick), max-norm and element-wise;
  3. the rollout cost of the candidate fixes is measured separately (bench.py under DVG_WINOGRAD=...).

Prints a markdown table (docs/DESIGN_NOTES_r06.md quotes it).  fused.WINOGRAD_LAYER_OVERRIDE forces the form per layer shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dvg_amd import fused  # noqa: E402
from oracle import params  # noqa: E402
from tests.common import rel_err, rel_err_elem, to64  # noqa: E402
from tests.test_gpu_configs import _build, _oracle_fns  # noqa: E402

DEV = "cuda:0"
# the eval-mode 3x3 layers of vgg_64 that run as F(4x4) at B = 50 / 64: (name, Cin, map side, Cout)
ENC_LAYERS = [("c2.0", 64, 32, 128), ("c2.1", 128, 32, 128), ("c3.0", 128, 16, 256), ("c3.1", 256, 16, 256), ("c3.2", 256, 16, 256),
              ("c4.0", 256, 8, 512), ("c4.1", 512, 8, 512), ("c4.2", 512, 8, 512)]
DEC_LAYERS = [("upc2.0 (x half)", 512, 8, 512), ("upc2.1", 512, 8, 512), ("upc2.2", 512, 8, 256), ("upc3.0 (x half)", 256, 16, 256),
              ("upc3.1", 256, 16, 256), ("upc3.2", 256, 16, 128), ("upc4.1", 128, 32, 64)]


def main():
    B = int(os.environ.get("DIAG_B", "50"))
    mods, (esd, dsd, lsd, gsd, lik) = _build("vgg", 64, 1, B, 2100)
    enc, dec = mods[0].to(DEV).eval(), mods[1].to(DEV).eval()
    x = params.frames(2110, B, 1, 64)
    vec = params.normal(2111, B, 90, scale=0.5).tanh()
    enc_o, dec_o = _oracle_fns("vgg", 64, esd, dsd)
    enc_6, dec_6 = _oracle_fns("vgg", 64, to64(esd), to64(dsd))
    with torch.no_grad():
        h32, sk32 = enc_o(x)
        y32 = dec_o(vec, sk32)
        h64, sk64 = enc_6(x.double())
        y64 = dec_6(vec.double(), sk64)
    xd, vd = x.to(DEV), vec.to(DEV)

    def run():
        fused.clear_skip_hoist_cache()
        from dvg_amd.rollout import drop_version_keyed_caches
        drop_version_keyed_caches()
        with torch.no_grad():
            h, sk = enc(xd)
            # the decoder on the ORACLE's skips (rounded to fp32): its own rounding, not the encoder's
            sko = [s.float().to(DEV).contiguous(memory_format=torch.channels_last) for s in sk64]
            fused.declare_frozen_skips(sko)      # the rollout's form: skip halves hoisted, x halves in Winograd form
            y = dec([vd, sko])
            y = dec([vd, sko])
        torch.cuda.synchronize()
        return h, y

    def figures(h, y):
        return (rel_err_elem(h, h64), rel_err(h, h64), rel_err_elem(y, y64), rel_err(y, y64))
    rows = []
    yard = figures(h32, y32)
    rows.append(("fp32 oracle (the reference's arithmetic)",) + yard)
    fused.WINOGRAD_LAYER_OVERRIDE.clear()
    base = figures(*run())
    rows.append(("HIP, default forms (F(4x4) on the layers below)",) + base)
    every = {(c, s, co): 0 for _, c, s, co in ENC_LAYERS + DEC_LAYERS}
    fused.WINOGRAD_LAYER_OVERRIDE.update(every)
    direct = figures(*run())
    rows.append(("HIP, every 3x3 layer direct",) + direct)
    f2 = {k: 2 for k in every}
    fused.WINOGRAD_LAYER_OVERRIDE.clear()
    fused.WINOGRAD_LAYER_OVERRIDE.update(f2)
    rows.append(("HIP, F(2x2) where F(4x4) ran (direct where F(2x2) does not exist)",) + figures(*run()))
    shapes = {}
    for name, c, s, co in ENC_LAYERS + DEC_LAYERS:      # layers of one shape share the override key: toggled together
        shapes.setdefault((c, s, co), []).append(name)
    for (c, s, co), names in shapes.items():
        name = " + ".join(names)
        for form, tag in ((0, "direct"), (2, "F(2x2)")):
            fused.WINOGRAD_LAYER_OVERRIDE.clear()
            fused.WINOGRAD_LAYER_OVERRIDE[(c, s, co)] = form
            rows.append((f"only {name} {c}->{co}@{s} {tag}, rest default",) + figures(*run()))
        fused.WINOGRAD_LAYER_OVERRIDE.clear()
        fused.WINOGRAD_LAYER_OVERRIDE.update(every)
        del fused.WINOGRAD_LAYER_OVERRIDE[(c, s, co)]
        rows.append((f"only {name} {c}->{co}@{s} F(4x4), rest direct",) + figures(*run()))
    fused.WINOGRAD_LAYER_OVERRIDE.clear()
    print(f"vgg_64 eval mode, B = {B}: error against the fp64 oracle (element-wise = max |a-b| / max(|b|, 1 % of max|b|))\n")
    print("| configuration | latent element-wise | latent max-norm | frame element-wise | frame max-norm |")
    print("|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | {r[1]:.2e} | {r[2]:.2e} | {r[3]:.2e} | {r[4]:.2e} |")
    print(f"\nratio HIP default / fp32 oracle: latent element-wise {base[0] / yard[0]:.2f}, frame element-wise {base[2] / yard[2]:.2f}; "
          f"all direct: {direct[0] / yard[0]:.2f} / {direct[2] / yard[2]:.2f}")


if __name__ == "__main__":
    main()
