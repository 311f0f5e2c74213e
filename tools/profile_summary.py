#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes) into
profiles/<tag>_<workload>_summary.{md,json}.  gfx950 correction (MI355X_MICROARCH.md §HBM):
FETCH_SIZE counts a wide coalesced read at half its bytes -> doubled; WRITE_SIZE is exact; both
are in KiB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void gs::", "").replace("gs::", "")
    return name.strip()


def main():
    out_dir, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    stats = {}
    for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            stats[short(row["Name"])] = dict(calls=int(row["Calls"]), total_ns=float(row["TotalDurationNs"]),
                                             avg_ns=float(row["AverageNs"]), pct=float(row["Percentage"]))
    pmc = defaultdict(lambda: defaultdict(list))
    for kind in ("fetch", "write"):
        for f in glob.glob(os.path.join(out_dir, "pmc_" + kind, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                pmc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    rows = []
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"]):
        fetch = pmc.get(k, {}).get("FETCH_SIZE")
        write = pmc.get(k, {}).get("WRITE_SIZE")
        fb = 2.0 * 1024.0 * sum(fetch) / len(fetch) if fetch else None
        wb = 1024.0 * sum(write) / len(write) if write else None
        rows.append(dict(kernel=k, calls=s["calls"], avg_us=s["avg_ns"] / 1e3, pct=s["pct"],
                         fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                         hbm_gbs=((fb or 0) + (wb or 0)) / s["avg_ns"] if (fb or wb) else None))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out", "profiles_out"), exist_ok=True)
    base = os.path.join(root, "gpurun_out", "profiles_out", "%s_%s_summary" % (tag, wl))
    json.dump(rows, open(base + ".json", "w"), indent=1)
    with open(base + ".md", "w") as f:
        f.write("# rocprofv3 summary — %s, workload %s\n\n" % (tag, wl))
        f.write("`rocprofv3 --kernel-trace --stats` (durations) + separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` "
                "passes; FETCH_SIZE doubled per the gfx950 correction, both KiB -> bytes.\n\n")
        f.write("| kernel | calls | avg us | % time | HBM fetch B/launch | HBM write B/launch | HBM GB/s |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %d | %.2f | %.1f | %s | %s | %s |\n" % (
                r["kernel"][:70], r["calls"], r["avg_us"], r["pct"],
                "%.3e" % r["fetch_bytes_per_launch"] if r["fetch_bytes_per_launch"] is not None else "-",
                "%.3e" % r["write_bytes_per_launch"] if r["write_bytes_per_launch"] is not None else "-",
                "%.0f" % r["hbm_gbs"] if r["hbm_gbs"] is not None else "-"))
    print(open(base + ".md").read())


if __name__ == "__main__":
    main()
