#!/usr/bin/env python3
"""Per-(kernel, grid size) durations of a rocprofv3 --kernel-trace CSV: inside a rollout the grid size identifies the layer
a launch belongs to, so this is the per-SHAPE view the --stats summary cannot give.  Analyses the longest burst of
back-to-back kernels (the graph replays).  Usage: trace_by_grid.py <kernel_trace.csv> [rollouts_in_burst]"""
import csv
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"(\w+)_kernel(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:50]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid // max(wg, 1)))
    rows.sort()
    bursts, cur, end = [], [rows[0]], rows[0][1]
    for r in rows[1:]:
        if r[0] - end > 200_000:
            bursts.append(cur)
            cur = []
        cur.append(r)
        end = max(end, r[1])
    bursts.append(cur)
    rows = max(bursts, key=len)
    nroll = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    span = rows[-1][1] - rows[0][0]
    agg = defaultdict(lambda: [0, 0])
    for s, e, n, g in rows:
        a = agg[(family(n), g)]
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print(f"burst: {len(rows)} kernels, span {span / 1e6:.3f} ms, kernel time {tot / 1e6:.3f} ms; per rollout (/{nroll:g}): "
          f"{span / 1e6 / nroll:.3f} / {tot / 1e6 / nroll:.3f} ms")
    for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t / tot < 0.002:
            continue
        print(f"  {k:44s} wgs {g:7d}  n/rollout {n / nroll:6.1f}  avg {t / n / 1e3:8.2f} us  per rollout {t / 1e6 / nroll:7.3f} ms  {100 * t / tot:5.1f} %")


if __name__ == "__main__":
    main()
