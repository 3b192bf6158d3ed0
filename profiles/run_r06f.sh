#!/bin/bash
# profiles/run_r06f.sh -- r06: checkpoint block choice (largest idle block instead of blind rotation) + the cluster run's block reserved once.
# Same-box A/B of bench_cluster.py 2 M (old behaviour: VSX_CK_ROTATE=1 VSX_CLUSTER_RESERVE=0), allpairs 20 k both ways, the whole -m gpu suite.
set -u
TAG=r06f
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
for V in old new old new; do
  E=""; [ $V = old ] && E="VSX_CK_ROTATE=1 VSX_CLUSTER_RESERVE=0"
  env $E VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 30000 > $OUT/cluster_$V.json 2> $OUT/cluster_$V.err
  echo "cluster $V rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/cluster_$V.json').read().strip().splitlines()[-1]); print(d['wall_s'], d['seconds_align_calls'], d['parity']['parity_sample_match'])")"
  grep -E "vsx_cluster_fast:" $OUT/cluster_$V.err | tail -1 | cut -c1-300
  grep "shared block" $OUT/cluster_$V.err | awk '{s+=$(NF-1); n++} END {print "   shared-block acquisitions: " n ", " s " ms"}'
done
for V in old new; do
  E=""; [ $V = old ] && E="VSX_CK_ROTATE=1"
  env $E timeout 600 python bench_allpairs.py --n 20000 --block 1000 --stream 1 > $OUT/allpairs20k_$V.json 2> $OUT/allpairs20k_$V.err
  echo "allpairs 20k $V rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/allpairs20k_$V.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['block_s'][:10], d['parity']['parity_sample_match'])" 2>&1 | cut -c1-400)"
done
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
echo "all done after $(el)"
