#!/usr/bin/env python3
"""Per-parameter gradient error of the train-mode vgg_64 / dcgan_64 encoder -> decoder backward against the fp64 oracle
(the case of tests/test_gpu_backward.py::test_module_backward_matches_reference_gradients), for the library that is loaded:
run once per build (DVG_HIP_LIB=...) and compare.  Prints max-entry and L2 errors, ours and the fp32 CPU oracle's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_backward import _oracle_grads, _reference_case, dev  # noqa: E402
from tests.test_oracle_golden import is_bn_fed_conv_bias  # noqa: E402


def main():
    family, seed = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("vgg", 210)
    enc, dec, esd, dsd, x, gy, gh = _reference_case(family, seed)
    h64, y64, e64, d64 = _oracle_grads(family, esd, dsd, x, gy, gh, torch.float64)
    h32, y32, e32, d32 = _oracle_grads(family, esd, dsd, x, gy, gh, torch.float32)
    enc.to(dev()).train(), dec.to(dev()).train()
    ho, so = enc(x.to(dev()))
    yo = dec([ho, so])
    ((yo * gy.to(dev())).sum() + (ho * gh.to(dev())).sum()).backward()
    print(f"forward: h {float((ho.double().cpu() - h64).abs().max() / h64.abs().max()):.2e} y {float((yo.double().cpu() - y64).abs().max() / y64.abs().max()):.2e}")
    rows = []
    for name, r64, r32, ours in (("enc", e64, e32, dict(enc.named_parameters())), ("dec", d64, d32, dict(dec.named_parameters()))):
        for k, p in ours.items():
            if is_bn_fed_conv_bias(k):
                continue
            g = r64[k].grad
            scale, norm = max(float(g.abs().max()), 1e-30), g.norm().clamp_min(1e-30)
            diff = p.grad.double().cpu() - g
            cdiff = r32[k].grad.double() - g
            rows.append((f"{name}.{k}", float(diff.abs().max()) / scale, float(diff.norm() / norm),
                         float(cdiff.abs().max()) / scale, float(cdiff.norm() / norm)))
    for r in rows:
        print(f"{r[0]:32s} max {r[1]:.2e} l2 {r[2]:.2e} | cpu fp32 max {r[3]:.2e} l2 {r[4]:.2e}")
    print(f"worst max {max(r[1] for r in rows):.2e} worst l2 {max(r[2] for r in rows):.2e} | mean l2 {sum(r[2] for r in rows) / len(rows):.2e} (cpu fp32 {sum(r[4] for r in rows) / len(rows):.2e})")


if __name__ == "__main__":
    main()
