#!/bin/bash
# profiles/ab_search.sh <tag> "<env>" ... -- the search_end_to_end leg of bench.py under different environments (8 calls each)
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
n=0
for E in "$@"; do
  n=$((n + 1))
  env $E VSX_BENCH_SEARCH_REPS=9 python bench.py --no-cpu --no-shapes --steps 1 --warmup 0 --e2e-calls 1 --ref-search-queries 0 > $OUT/run$n.json 2> $OUT/run$n.err
  python - $OUT/run$n.json "$E" <<'PY'
import json, sys
try:
    s = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["search_end_to_end"]
    print(f"{sys.argv[2]:34s} | best {s['seconds']} s = {s['queries_per_s']} q/s | median {s['seconds_median']} | calls {s['seconds_later_calls']}")
except Exception as ex:
    print(sys.argv[2], "unreadable:", ex)
PY
done
