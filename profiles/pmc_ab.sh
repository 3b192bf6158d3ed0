#!/bin/bash
# profiles/pmc_ab.sh <library> -- FETCH_SIZE / WRITE_SIZE / SQ pass of one kernels-only bench step for an A/B build of libvsx
# (VSX_LIBRARY); prints per-kernel sums.  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
export VSX_LIBRARY=$1
shift
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY"; do
  rm -rf /tmp/pq; rocprofv3 --output-format csv --kernel-trace --pmc $C -d /tmp/pq -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --kernels-only "$@" > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:44]
        if "vsx_f" in k or "vsx_t" in k: agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
for k,v in agg.items(): print(k, {c: "%.4g"%x for c,x in v.items()})
PY
done
