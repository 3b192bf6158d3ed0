#!/bin/bash
# profiles/run_r06u.sh -- r06: where config 5's per-GPU share (1.25 M x 150 bp vs 5 M x 1 kbp) spends a search call: per-window timeline + kernel totals
set -u
TAG=r06u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd /tmp && export TMPDIR=/tmp
ARGS="--queries 1250000 --qlen 150 --db 5000000 --dlen 1000 --no-cpu --no-shapes --steps 1 --warmup 0 --e2e-calls 1 --ref-search-queries 0"
VSX_BENCH_SEARCH_REPS=2 VSX_DEBUG_TIMELINE=1 VSX_DEBUG_TIMING=1 rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/c5.json 2> $OUT/c5.err
for f in $(find $WORK/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
python - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-80s calls=%-6s total_ms=%9.1f avg_us=%9.1f" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
grep -E "vsx_search_batch:" $OUT/c5.err | tail -2
grep -E "ms\]" $OUT/c5.err | tail -260 > $OUT/timeline.txt; head -30 $OUT/timeline.txt; echo ...; tail -40 $OUT/timeline.txt
