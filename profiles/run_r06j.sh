#!/bin/bash
# profiles/run_r06j.sh -- r06: allpairs 50 k on the current build + a 3-block timeline
set -u
TAG=r06j
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python bench_allpairs.py --n 50000 --block 1000 --stream 1 > $OUT/bench_allpairs_50k.json 2> $OUT/bench_allpairs_50k.err
echo "allpairs 50k rc=$?: $(python -c "import json; d=json.loads(open('$OUT/bench_allpairs_50k.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:12], d['parity'])" 2>&1 | cut -c1-700)"
VSX_DEBUG_TIMING=1 timeout 300 python bench_allpairs.py --n 50000 --block 1000 --stream 1 --max-blocks 3 --parity-prefix 0 > $OUT/allpairs_3blocks.json 2> $OUT/allpairs_3blocks.err
grep -vE "shared block" $OUT/allpairs_3blocks.err | tail -150 | cut -c1-220 > $OUT/allpairs_3blocks_tail.txt
tail -60 $OUT/allpairs_3blocks_tail.txt
