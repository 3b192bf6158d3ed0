#!/bin/bash
# profiles/run_r05n.sh -- round 5: --cluster_fast 2 M, lazy first batches on / off / on / off in one call (order reversed against r05m), then the cluster soak with it on
set -u
TAG=r05n
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
for LZ in 1 0 1 0; do
  VSX_CLUSTER_LAZY=$LZ VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 0 > $OUT/cluster_lazy$LZ.json 2> $OUT/cluster_lazy$LZ.err
  echo "cluster lazy=$LZ after $(( $(date +%s) - T0 )) s: wall $(python -c "import json; d=json.loads(open('$OUT/cluster_lazy$LZ.json').read().strip().splitlines()[-1]); print(d['wall_s'], d['stages'], d['pairs_aligned'])")"
  grep -E "vsx_cluster_fast:" $OUT/cluster_lazy$LZ.err | tail -1 | cut -c1-400
done
VSX_CLUSTER_LAZY=1 timeout 120 python oracle/soak_cluster.py --seconds 45 --seed 20260930 --out gpurun_out/$TAG/soak_cluster.json > $OUT/soak_cluster.log 2>&1
echo "soak_cluster (lazy) rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/soak_cluster.json')); print({k: v for k, v in d.items() if k not in ('failures', 'examples', 'what')})" 2>&1 | cut -c1-300)"
