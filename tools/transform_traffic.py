#!/usr/bin/env python3
"""profiles/rRR_transform_traffic.json: for every Winograd transform / hand-over kernel of the vgg_64 rollout, the HBM-side
bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, profiles/rRR_pmc_by_kernel.json), its average duration from
the kernel-trace statistics of the single-chain run (profiles/rRR_vgg64_rollout_kernel_stats.csv), the rate the two give and
the LDS bank-conflict share.   usage: tools/transform_traffic.py 05"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name):
    m = re.search(r"(\w+)_kernel(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    rr = sys.argv[1]
    pmc = json.load(open(os.path.join(ROOT, "profiles", f"r{rr}_pmc_by_kernel.json")))["vgg"]
    stats = {}
    for row in csv.DictReader(open(os.path.join(ROOT, "profiles", f"r{rr}_vgg64_rollout_kernel_stats.csv"))):
        stats[family(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6)
    out, tot_ms = {}, 0.0
    for k, c in sorted(pmc.items()):
        if not (k.startswith("winograd") or k.startswith("stem_up")) or "weight" in k or "FETCH_SIZE" not in c:
            continue
        traffic = (2.0 * c["FETCH_SIZE"]["avg"] + c["WRITE_SIZE"]["avg"]) * 1024
        st = stats.get(k)
        e = {"fetch_kb_raw": round(c["FETCH_SIZE"]["avg"], 1), "write_kb": round(c["WRITE_SIZE"]["avg"], 1),
             "traffic_bytes_per_launch": int(traffic)}
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]["avg"] > 0:
            e["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"]["avg"] / c["SQ_LDS_IDX_ACTIVE"]["avg"], 3)
        if st:
            e.update(calls_in_stats_run=st[0], avg_us=round(st[1], 2), total_ms_in_stats_run=round(st[2], 2),
                     counter_gbs=round(traffic / (st[1] * 1e-6) / 1e9, 1), counter_frac_of_hbm_peak=round(traffic / (st[1] * 1e-6) / 8e12, 3))
            tot_ms += st[2]
        out[k] = e
    out["_note"] = ("bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction, MI355X_MICROARCH.md), averaged over the "
                    "launches of the PMC passes (eager, --steps 2); durations = rocprofv3 --kernel-trace --stats of the hipGraph "
                    "single-chain run; launches at B = 64 and at the conditioning batch B = 576 are averaged together")
    out["_transform_ms_in_stats_run"] = round(tot_ms, 2)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"r{rr}_transform_traffic.json"), "w"), indent=1)
    for k, e in out.items():
        print(k, e)


if __name__ == "__main__":
    main()
