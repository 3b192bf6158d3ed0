#!/bin/bash
# profiles/run_r06i.sh -- r06: host passes of search / clustering on the persistent worker pool; cluster 2 M x3, search x9, suite
set -u
TAG=r06i
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for V in 1 2 3; do
  VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 100000 > $OUT/cluster_$V.json 2> $OUT/cluster_$V.err
  echo "cluster $V rc=$?: $(python -c "import json; d=json.loads(open('$OUT/cluster_$V.json').read().strip().splitlines()[-1]); print(d['wall_s'], d['seconds_align_calls'], d['clusters'], d['parity']['parity_sample_match'])")"
  grep -E "vsx_cluster_fast:" $OUT/cluster_$V.err | tail -1 | cut -c1-330
done
bash profiles/ab_search.sh $TAG/ab "VSX_X=0" "VSX_X=1"
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
