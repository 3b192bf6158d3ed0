#!/bin/bash
# profiles/run_r06d.sh -- r06: a window's hits in one vector (no 10^5 small blocks released at return); search tests
set -u
TAG=r06d
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
VSX_DEBUG_TIMELINE=1 VSX_DEBUG_TIMING=1 VSX_BENCH_SEARCH_REPS=6 python bench.py --no-cpu --no-shapes --steps 1 --warmup 0 --e2e-calls 5 --ref-search-queries 0 \
    > $OUT/search_timeline.json 2> $OUT/search_timeline.err
grep -E "vsx_search_batch:" $OUT/search_timeline.err | tail -6
python - $OUT/search_timeline.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["search_end_to_end"]
print({k: s.get(k) for k in ("queries_per_s", "queries_per_s_best", "seconds_later_calls", "cpu_throttle_during_later_calls")})
e = d["end_to_end"]
print({k: e.get(k) for k in ("value", "value_median", "ms_calls", "cpu_throttle_during_calls")})
PY
bash profiles/ab_search.sh $TAG/ab "VSX_X=0" "VSX_SEARCH_TAPER=3" "VSX_X=1"
timeout 900 python -m pytest tests -x -q -m gpu -k "search or filters or mask or scale" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
