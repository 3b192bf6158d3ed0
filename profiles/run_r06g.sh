#!/bin/bash
# profiles/run_r06g.sh -- r06: where a --cluster_fast 2 M run spends the device: rocprofv3 kernel trace (stats per kernel) of bench_cluster.py
set -u
TAG=r06g
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd /tmp && export TMPDIR=/tmp
VSX_DEBUG_TIMING=1 rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- python $REPO/bench_cluster.py --n 2000000 --parity-prefix 0 > $OUT/cluster_trace.json 2> $OUT/cluster_trace.err
for f in $(find $WORK/trace -name "*kernel_stats.csv"); do cp $f $OUT/cluster_kernel_stats.csv; done
python - $OUT/cluster_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("summed kernel time %.3f s" % (tot / 1e9))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-90s calls=%-7s total_ms=%9.1f avg_us=%9.1f" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
grep -E "vsx_cluster_fast:" $OUT/cluster_trace.err | tail -1 | cut -c1-300
tail -c 600 $OUT/cluster_trace.json
