#!/bin/bash
# profiles/run_r06c.sh -- r06: search call, host state released on the marshal threads; CPU throttling counters around the warm calls
set -u
TAG=r06c
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
cat /sys/fs/cgroup/cpu.stat > $OUT/cpu_stat_before.txt 2>&1
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
cat /sys/fs/cgroup/cpu.stat > $OUT/cpu_stat_after.txt 2>&1
bash profiles/ab_search.sh $TAG/ab "VSX_X=0" "VSX_SEARCH_THREADS=8" "VSX_SEARCH_THREADS=12" "VSX_SEARCH_CONSUMERS=2"
