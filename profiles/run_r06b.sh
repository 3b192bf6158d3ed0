#!/bin/bash
# profiles/run_r06b.sh -- r06: the purity pass deferred (no device-wide sync per window plan): search timeline + 9-call figure
set -u
TAG=r06b
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
VSX_DEBUG_TIMELINE=1 VSX_DEBUG_TIMING=1 VSX_BENCH_SEARCH_REPS=5 python bench.py --no-cpu --no-shapes --steps 1 --warmup 0 --e2e-calls 1 --ref-search-queries 0 \
    > $OUT/search_timeline.json 2> $OUT/search_timeline.err
grep -E "vsx_search_batch:" $OUT/search_timeline.err | tail -3
grep -n "ms\]" $OUT/search_timeline.err | tail -75 | cut -c1-200
bash profiles/ab_search.sh $TAG/ab "VSX_X=0" "VSX_SEARCH_CONSUMERS=2" "VSX_SEARCH_RANKERS=2" "VSX_SEARCH_RANKERS=4"
timeout 900 python -m pytest tests -x -q -m gpu -k "pair_profile or sparse or search or parity" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
