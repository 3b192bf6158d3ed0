#!/bin/bash
# profiles/run_r06k.sh -- r06: ranked plans compact their kept pairs in the traceback's epilogue (no flag kernel / rocPRIM scan / segmented sort
# over all pairs).  Tests that touch the ranked path, then allpairs 20 k A/B (VSX_RANK_PRIM=1 = the r05 path), then 50 k.
set -u
TAG=r06k
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -x -q -m gpu -k "allpairs or rank or filter or search or soak" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for V in prim lists prim lists; do
  E=""; [ $V = prim ] && E="VSX_RANK_PRIM=1"
  env $E timeout 600 python bench_allpairs.py --n 20000 --block 1000 --stream 1 > $OUT/allpairs20k_$V.json 2> $OUT/allpairs20k_$V.err
  echo "allpairs 20k $V rc=$?: $(python -c "import json; d=json.loads(open('$OUT/allpairs20k_$V.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['accepted_hits'], d['block_s'][:10], d['parity']['parity_sample_match'])" 2>&1 | cut -c1-400)"
done
timeout 900 python bench_allpairs.py --n 50000 --block 1000 --stream 1 > $OUT/bench_allpairs_50k.json 2> $OUT/bench_allpairs_50k.err
echo "allpairs 50k rc=$?: $(python -c "import json; d=json.loads(open('$OUT/bench_allpairs_50k.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:12], d['parity'])" 2>&1 | cut -c1-700)"
