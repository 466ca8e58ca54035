#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/profile_round.sh: per-kernel-family averages of the PMC counters (one CSV
row per dispatch x counter) and the kernel-trace stats.  Usage: pmc_summary.py gpurun_out/prof_round [dir-glob, default pmc_*]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"(\w+)_kernel(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    pat = sys.argv[2] if len(sys.argv) > 2 else "pmc_*"
    for f in glob.glob(os.path.join(root, pat, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[family(row["Kernel_Name"])][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    out = {}
    for fam, cs in sorted(acc.items()):
        out[fam] = {c: {"avg": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
