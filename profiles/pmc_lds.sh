#!/bin/bash
# profiles/pmc_lds.sh -- LDS counters of one bench step (GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl; rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU -d /tmp/pl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pl/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:40]
        if "vsx_f" in k or "vsx_t" in k: agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
for k,v in agg.items(): print(k, {c: "%.4g"%x for c,x in v.items()})
PY
