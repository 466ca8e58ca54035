#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: per-kernel-family totals plus how much of the wall-clock span the GPU was
idle between kernels (launch gaps).  Usage: trace_gaps.py <kernel_trace.csv> [t_start_frac]  (analyses the part of the
trace after t_start_frac of its span, default 0.5 = the steady-state replays)."""
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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    arg = sys.argv[2] if len(sys.argv) > 2 else "burst"
    if arg == "burst":          # the longest run of kernels without an idle gap > 200 us (back-to-back graph replays)
        bursts, cur = [], [rows[0]]
        end = rows[0][1]
        for r in rows[1:]:
            if r[0] - end > 200_000:
                bursts.append(cur)
                cur = []
            cur.append(r)
            end = max(end, r[1])
        bursts.append(cur)
        rows = max(bursts, key=len)
    else:
        t0 = rows[0][0] + float(arg) * (rows[-1][1] - rows[0][0])
        rows = [r for r in rows if r[0] >= t0]
    span = rows[-1][1] - rows[0][0]
    busy, cur_end, gaps, overlap = 0, rows[0][0], [], 0
    fam = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        f = fam[family(n)]
        f[0] += 1
        f[1] += e - s
        if s > cur_end:
            gaps.append(s - cur_end)
            busy += e - s
            cur_end = e
        else:
            overlap += min(e, cur_end) - s
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
    print(f"span {span / 1e3:.1f} us, union-busy {busy / 1e3:.1f} us ({100 * busy / span:.1f} %), kernels {len(rows)}, "
          f"gaps {len(gaps)} totalling {sum(gaps) / 1e3:.1f} us (median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us), "
          f"overlapped kernel time {overlap / 1e3:.1f} us")
    for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:40s} n={n:5d} total {t / 1e3:9.1f} us avg {t / n / 1e3:7.2f} us  {100 * t / span:5.1f} % of span")


if __name__ == "__main__":
    main()
