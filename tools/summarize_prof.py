#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output to small per-kernel summaries (run on the GPU box by tools/profile.sh)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0][:60]


def trace(d, out):
    rows = defaultdict(list)
    for f in find(d, "*kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in rows.values()) or 1.0
    with open(out, "w") as o:
        o.write("%-62s %8s %12s %12s %12s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            o.write("%-62s %8d %12.1f %12.1f %12.1f %12.1f %7.2f\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))


def pmc(d, out):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in find(d, "*counter_collection.csv"):
        with open(f) as fh:
            seen = set()
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (r.get("Dispatch_Id"), k)
                if key not in seen:
                    seen.add(key)
                    cnt[k] += 1
    with open(out, "w") as o:
        o.write("per-kernel counter sums over all dispatches (and per-dispatch average)\n")
        for k in sorted(acc):
            for c, v in sorted(acc[k].items()):
                o.write("%-62s dispatches=%-6d %-36s sum=%.6g avg=%.6g\n" % (k, cnt[k], c, v, v / max(cnt[k], 1)))


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
