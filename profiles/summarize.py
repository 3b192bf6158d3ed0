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


# ---- HBM traffic of the DP kernel per launch (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are KiB collected in
# separate passes; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (doubled here); WRITE_SIZE was
# calibrated on vsx_encode_kernel in the same run (it writes exactly one byte per input byte: see its row above).
import json


def per_launch(tag, counter, needle):
    tot, n = 0.0, set()
    for f in find(f"{tag}/**/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if needle in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    tot += float(row.get("Counter_Value", 0) or 0)
                    n.add(row.get("Dispatch_Id"))
    return (tot / len(n)) if n else None


fetch = per_launch("pmc_fetch", "FETCH_SIZE", "vsx_forward_kernel")
write = per_launch("pmc_write", "WRITE_SIZE", "vsx_forward_kernel")
if fetch is not None and write is not None:
    doc = {"kernel": "vsx_forward_kernel", "fetch_size_kib": fetch, "write_size_kib": write,
           "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
           "workload": {"queries": 100000, "qlen": 250, "db": 1000000, "dlen": 1000, "cands": 8},
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 0 "
                     "--no-cpu`; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"}
    with open(os.path.join(os.path.dirname(out.rstrip('/')) if False else out, "traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    print("== traffic ==")
    print(json.dumps(doc))
