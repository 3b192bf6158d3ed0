#!/bin/bash
# profiles/run_r06h.sh -- r06: cluster_fast with the round state released on a helper thread + record-form hits; A/B VSX_CLUSTER_REAPER=0; round sizes
set -u
TAG=r06h
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for V in off on off on; do
  E=""; [ $V = off ] && E="VSX_CLUSTER_REAPER=0"
  env $E VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 100000 > $OUT/cluster_$V.json 2> $OUT/cluster_$V.err
  echo "cluster reaper $V rc=$?: $(python -c "import json; d=json.loads(open('$OUT/cluster_$V.json').read().strip().splitlines()[-1]); print(d['wall_s'], d['seconds_align_calls'], d['clusters'], d['parity']['parity_sample_match'])")"
  grep -E "vsx_cluster_fast:" $OUT/cluster_$V.err | tail -1 | cut -c1-330
done
VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 0 --round 32768 > $OUT/cluster_r32k.json 2> $OUT/cluster_r32k.err
echo "cluster round 32768: $(python -c "import json; d=json.loads(open('$OUT/cluster_r32k.json').read().strip().splitlines()[-1]); print(d['wall_s'], d['seconds_align_calls'], d['clusters'])")"
grep -E "vsx_cluster_fast:" $OUT/cluster_r32k.err | tail -1 | cut -c1-330
timeout 900 python -m pytest tests -x -q -m gpu -k "cluster or scale or soak or api or shim" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
