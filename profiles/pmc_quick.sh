#!/bin/bash
# profiles/pmc_quick.sh -- one SQ counter pass over a single bench step (run on the GPU box); prints per-kernel sums
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq; rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/pq -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:40]
        if "vsx_f" in k or "vsx_t" in k: agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
for k,v in agg.items(): print(k, {c: "%.4g"%x for c,x in v.items()})
PY
