#!/bin/bash
# profiles/run_r06s_long.sh -- r06: long soaks on the final build (fresh seed): the worker pool, the reaper, the block choice, the ranking lists and
# the ONE kernels under randomized options for minutes instead of 40 s
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/r06long; mkdir -p $OUT; cd $REPO
for spec in "soak_cluster 300" "soak_search 300" "soak_allpairs 200" "soak 200" "soak_api 120"; do
  set -- $spec
  python oracle/$1.py --seconds $2 --seed 66006 --out $OUT/$1.json > $OUT/$1.log 2>&1
  echo "$1 rc=$? $(python -c "import json; d=json.load(open('$OUT/$1.json')); print({k: v for k, v in d.items() if k not in ('failures', 'examples', 'what', 'shapes', 'scoring_kinds', 'by_command')})" 2>&1 | cut -c1-300)"
done
