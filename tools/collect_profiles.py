#!/usr/bin/env python3
"""Turn gpurun_out/prof_round/ (tools/profile_round.sh) into the committed evidence under profiles/ for round RR:
   python tools/collect_profiles.py 02"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
DST = os.path.join(ROOT, "profiles")


def family(name):
    m = re.search(r"(\w+)_kernel(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def pmc(pattern):
    acc = {}
    for f in glob.glob(os.path.join(SRC, pattern, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc.setdefault(family(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: {c: {"avg": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()} for k, cs in sorted(acc.items())}


def main():
    rr = sys.argv[1]
    tag = f"r{rr}"
    for sub, name in (("stats_vgg", "vgg64_rollout"), ("stats_dcgan", "dcgan64_rollout"), ("stats_train", "train_vgg64"),
                      ("stats_train_dcgan", "train_dcgan64"), ("stats_vgg_inflight3", "vgg64_rollout_inflight3"),
                      ("stats_dcgan_inflight3", "dcgan64_rollout_inflight3")):
        f = glob.glob(os.path.join(SRC, sub, "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(DST, f"{tag}_{name}_kernel_stats.csv"))
    for f, name in (("bench.json", "bench.json"), ("train_graph.jsonl", "train_graphed.jsonl")):
        if os.path.exists(os.path.join(SRC, f)):
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, f"{tag}_{name}"))
    # the bench lines printed by the profiled commands themselves (their live-event roofline legs ran under rocprofv3)
    for m in ("vgg", "dcgan"):
        f = os.path.join(SRC, f"bench_{m}_under_rocprof.log")
        if os.path.exists(f):
            lines = [ln for ln in open(f) if ln.lstrip().startswith("{") and '"metric"' in ln]
            if lines:
                open(os.path.join(DST, f"{tag}_{m}64_rollout_bench_under_rocprof.json"), "w").write(lines[-1])
    out = {}
    for m in ("vgg", "dcgan"):
        out[m] = pmc(f"pmc_{m}_*")
    # HBM-side bytes of ONE rollout: sum over every kernel of (2 x FETCH_SIZE + WRITE_SIZE) KB x its dispatches, divided by the
    # rollouts the profiled command ran (= dispatches of the GP sampling kernel: one trigger step per 10-in/10-out rollout)
    for m in ("vgg", "dcgan"):
        ks = out[m]
        n_roll = max([c["FETCH_SIZE"]["dispatches"] for k, c in ks.items() if k.startswith("gp_predict") and "FETCH_SIZE" in c] or [0])
        if not n_roll:
            continue
        tot = sum((2.0 * c["FETCH_SIZE"]["avg"] * c["FETCH_SIZE"]["dispatches"] + c["WRITE_SIZE"]["avg"] * c["WRITE_SIZE"]["dispatches"])
                  for c in ks.values() if "FETCH_SIZE" in c and "WRITE_SIZE" in c) * 1024
        ks["_rollout"] = {"traffic_bytes_per_rollout": tot / n_roll, "rollouts": n_roll,
                          "note": "2 x FETCH_SIZE + WRITE_SIZE over all kernels of the profiled command / its rollouts"}
    json.dump(out, open(os.path.join(DST, f"{tag}_pmc_by_kernel.json"), "w"), indent=1)
    # HBM traffic of the dominant kernels per launch: FETCH_SIZE (KB; x2 on gfx950 for wide streaming reads, the
    # MI355X_MICROARCH.md correction) + WRITE_SIZE (KB), launch-weighted over the instantiations of a family
    fams = {"conv3x3": ("vgg", [k for k in out["vgg"] if k.startswith("conv_igemm2<0")]),
            "winograd_gemm": ("vgg", [k for k in out["vgg"] if k.startswith("conv_igemm2<3")]),
            "conv4x4s2": ("dcgan", [k for k in out["dcgan"] if k.startswith("conv_igemm2<1")]),
            "convT4x4s2": ("dcgan", [k for k in out["dcgan"] if k.startswith("conv_igemm2<2")])}
    for name, (m, keys) in fams.items():
        n = fetch = write = 0.0
        for k in keys:
            c = out[m][k]
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            d = c["FETCH_SIZE"]["dispatches"]
            n += d
            fetch += c["FETCH_SIZE"]["avg"] * d
            write += c["WRITE_SIZE"]["avg"] * d
        if n:
            json.dump({"kernel": name if name.startswith("winograd") else f"{name}_igemm", "family": m, "dispatches": int(n),
                       "fetch_kb_per_launch_raw": fetch / n, "write_kb_per_launch": write / n,
                       "traffic_bytes_per_launch": int((2.0 * fetch / n + write / n) * 1024),
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --model %s --steps 2 "
                               "--warmup 1 --no-graph`; FETCH_SIZE doubled (gfx950: wide streaming reads are reported at half)" % m},
                      open(os.path.join(DST, f"{tag}_{name}_traffic.json"), "w"), indent=1)
    print(subprocess.run(["ls", "-la", DST], capture_output=True, text=True).stdout)


if __name__ == "__main__":
    main()
