#!/usr/bin/env python3
"""profiles/pmc_kmer.py <workdir> <out.json> -- the counting kernel's PMC results of `bench_kmer.py --host-queries 0 --repeat 1`
(separate rocprofv3 passes under <workdir>/pmc_fetch, pmc_write, pmc_sq; profiles/run_r06z.sh) -> profiles/pmc_kmer_current.json, the
file bench_kmer.py reads for roofline.traffic (it refuses it once the k-mer kernel sources change)."""
import csv
import glob
import json
import os
import sys

work, dst = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_kmer  # noqa: E402


def total(tag, counter, needle="vsx_kmer_count"):
    tot, n = 0.0, set()
    for f in glob.glob(os.path.join(work, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if needle in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                tot += float(row.get("Counter_Value", 0) or 0)
                n.add(row.get("Dispatch_Id"))
    return tot, len(n)


fetch, nd = total("pmc_fetch", "FETCH_SIZE")
write, _ = total("pmc_write", "WRITE_SIZE")
sq = {c: total("pmc_sq", c)[0] for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                         "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT")}
doc = {"kernel_source_sha": bench_kmer.kmer_source_sha(), "kernel_sources": list(bench_kmer.KMER_SOURCES),
       "workload": {"queries": 100000, "qlen": 250, "db": 1000000, "dlen": 1000},
       "count_kernel": {"fetch_size_kib": fetch, "write_size_kib": write, "dispatches": nd,
                        "hbm_bytes_per_batch": int((2 * fetch + write) * 1024), "sq": sq}}
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(doc))
