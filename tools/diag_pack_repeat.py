#!/usr/bin/env python3
"""DIAGNOSTIC (r06, profiles/r06_dp_race_bisect.txt): the per-weight-version cache entries of the training path (packed igemm
weights, Winograd-domain weights of the forward and of the data gradient, 2-D transposes) recomputed `--iters` times each while
ANOTHER PROCESS keeps the device busy (tools/diag_repeat_backward.py --noise_child train | nan), every result compared bit for bit
with the first.  tools/diag_repeat_backward.py --drop_caches --noise proctrain showed that recomputing these entries under
cross-process contention changes the gradients; this tool says which entry."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--explain", type=int, default=0, help="describe the differing rows of the first N differing recomputations per entry")
    ap.add_argument("--noise", default="train", choices=["none", "train", "nan", "mm"])
    a = ap.parse_args()
    import torch
    from dvg_amd import ops
    dev = torch.device("cuda", 0)
    child = None
    if a.noise != "none":
        cmd = [sys.executable, os.path.join(ROOT, "tools", "diag_repeat_backward.py"), "--noise_child", a.noise, "--model", "vgg",
               "--batch", "4", "--repeats", "100000"]
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        assert child.stdout.readline().strip() == "ready"
    g = torch.Generator(device="cpu").manual_seed(3)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(dev)
    cases = []
    for co, ci in ((64, 64), (128, 64), (128, 128), (256, 128), (256, 256), (512, 256), (512, 512), (512, 1024), (256, 512)):
        w = rnd(co, ci, 3, 3)
        cases.append((f"winograd_weight m=4 {co}x{ci}", lambda w=w: ops.winograd_weight(w, 4)))
        cases.append((f"winograd_weight m=2 {co}x{ci}", lambda w=w: ops.winograd_weight(w, 2)))
        cases.append((f"winograd_weight m=4 dgrad-form {co}x{ci}",
                      lambda w=w: ops.winograd_weight(w.transpose(0, 1).flip(2, 3).contiguous(), 4)))
        cases.append((f"pack_igemm 3x3 {co}x{ci}", lambda w=w: ops.pack_igemm_weight(w, False)))
        cases.append((f"pack_igemm 3x3 transposed {co}x{ci}", lambda w=w: ops.pack_igemm_weight(w, True) if ci % 64 == 0 else None))
    for co, ci in ((64, 64), (128, 64), (256, 128), (512, 256)):
        w4 = rnd(co, ci, 4, 4)
        cases.append((f"pack_igemm 4x4 {co}x{ci}", lambda w=w4: ops.pack_igemm_weight(w, False)))
        cases.append((f"pack_igemm 4x4 transposed {co}x{ci}", lambda w=w4: ops.pack_igemm_weight(w, True)))
    for r, c in ((1024, 256), (256, 1024), (90, 256), (346, 1024)):
        m = rnd(r, c)
        cases.append((f"transpose2d {r}x{c}", lambda m=m: ops.transpose2d(m)))
    torch.cuda.synchronize()
    bad_total = 0
    for name, fn in cases:
        first = fn()
        if first is None:
            continue
        first = first.clone()
        bad, worst, explained = 0, 0, 0
        for _ in range(a.iters):
            out = fn()
            if not torch.equal(out, first):
                bad += 1
                ne = (out.view(torch.int32) != first.view(torch.int32)).reshape(-1)
                worst = max(worst, int(ne.sum()))
                if a.explain and explained < a.explain and name.startswith("winograd_weight m=4"):
                    explained += 1
                    o32, f32 = out.view(torch.int32).reshape(-1, 24), first.view(torch.int32).reshape(-1, 24)
                    rows = torch.nonzero(ne.reshape(-1, 24).any(dim=1)).reshape(-1).tolist()
                    base = out.data_ptr()
                    # every reference row by content, to recognise a row that landed in the wrong place
                    ref = {bytes(r.cpu().numpy().tobytes()): i for i, r in enumerate(f32)} if f32.shape[0] <= 200000 else {}
                    for r_ in rows[:12]:
                        got = o32[r_]
                        kind = "zeros" if int(got.abs().sum()) == 0 else ("= reference row %d" % ref[bytes(got.cpu().numpy().tobytes())]
                                                                          if bytes(got.cpu().numpy().tobytes()) in ref else "other")
                        nbad = int((got != f32[r_]).sum())
                        addr = base + r_ * 96
                        print(f"   {name}: row {r_} (co%64 {r_ % 64}, block {r_ // 64}) {nbad}/24 words wrong, content {kind}; "
                              f"address {addr:#x} (mod 128: {addr % 128}, mod 4096: {addr % 4096}, mod 2 MiB: {addr % (1 << 21):#x})",
                              flush=True)
        torch.cuda.synchronize()
        bad_total += bad
        if bad:
            print(f"{name}: {bad} of {a.iters} recomputations differ (up to {worst} of {first.numel()} words)", flush=True)
    if child is not None:
        child.terminate()
        child.wait()
    print(f"diag_pack_repeat: noise {a.noise}: {bad_total} differing recomputations over {len(cases)} entries x {a.iters}", flush=True)


if __name__ == "__main__":
    main()
