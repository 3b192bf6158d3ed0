#!/bin/bash
# profiles/run_r05r.sh -- round 5: allpairs at 20 000 sequences with the DP / traceback overlap of neighbouring slices on (default) and off (VSX_ALIGN_SERIAL=1)
set -u
TAG=r05r
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for SR in 0 1 0 1; do
  VSX_ALIGN_SERIAL=$SR timeout 300 python bench_allpairs.py --n 20000 --block 1000 --stream 1 --parity-prefix 0 > $OUT/allpairs_serial$SR.json 2> $OUT/allpairs_serial$SR.err
  echo "serial=$SR: $(python -c "import json; d=json.loads(open('$OUT/allpairs_serial$SR.json').read().strip().splitlines()[-1]); print(d['wall_s'], round(sum(d['block_s'][1:]), 3), d['block_s'][:8])" 2>&1 | cut -c1-300)"
done
