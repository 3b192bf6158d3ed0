#!/bin/bash
# profiles/run_r05q.sh -- round 5: where the GPU time of --allpairs_global goes: rocprofv3 kernel trace of the first two 1 000-query blocks of the 50 000 run
set -u
TAG=r05q
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- python $REPO/bench_allpairs.py --n 50000 --block 1000 --max-blocks 2 --parity-prefix 0 > $OUT/trace.log 2>&1
for f in $(find $WORK/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
python - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.1f} ms avg {float(r['AverageNs'])/1e6:8.3f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} %")
print("sum of kernel time", round(tot / 1e6, 1), "ms")
PY
tail -2 $OUT/trace.log | cut -c1-400
