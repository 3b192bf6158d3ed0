#!/bin/bash
# profiles/run_r05g.sh -- round 5, seventh GPU call, before the final records: the six randomized soaks against the reference on a fresh
# seed (the planner, the DP kernel's sparse classes, the traceback's LDS staging, the shim and allpairs all changed this round), the
# allpairs tests incl. the stream, and the stream A/B at 20 000 sequences again.  Under gpurun_out/r05g/.
set -u
TAG=r05g
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_scale.py tests/test_gpu_shim.py -x -q -k "allpairs or shim" > $OUT/tests.log 2>&1
echo "allpairs + shim tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head
bash profiles/run_soaks.sh $TAG 45 20260926 2>&1 | tee $OUT/soaks.txt
echo "soaks done after $(el)"
for ST in 0 1; do
  timeout 600 python bench_allpairs.py --n 20000 --block 1000 --stream $ST --parity-prefix 0 > $OUT/allpairs_20k_stream$ST.json 2> $OUT/allpairs_20k_stream$ST.err
  echo "allpairs 20k stream=$ST rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/allpairs_20k_stream$ST.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:6])" 2>&1 | cut -c1-300)"
done
echo "all done after $(el)"
