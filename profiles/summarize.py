#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory: kernel-trace stats + PMC counters per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")[:70]
            print(f"{name:70s} calls={row.get('Calls')} total_ns={row.get('TotalDurationNs')} "
                  f"avg_ns={row.get('AverageNs')} pct={row.get('Percentage')}")

for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    files = find(f"{tag}/**/*counter_collection.csv")
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    seen = set()
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")[:60]
                c = row.get("Counter_Name")
                v = float(row.get("Counter_Value", 0) or 0)
                agg[k][c] += v
                key = (k, row.get("Dispatch_Id"))
                if key not in seen:
                    seen.add(key)
                    cnt[k] += 1
    print(f"== {tag} (sum over dispatches; n = dispatches) ==")
    for k, cs in agg.items():
        print(f"{k:60s} n={cnt[k]} " + " ".join(f"{c}={v:.6g}" for c, v in sorted(cs.items())))
