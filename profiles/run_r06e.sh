#!/bin/bash
# profiles/run_r06e.sh -- r06: the secondary commands after the deferred purity pass (no device-wide sync per plan): cluster_fast 2 M, allpairs 50 k, shim cost
set -u
TAG=r06e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 100000 > $OUT/bench_cluster_2M.json 2> $OUT/bench_cluster_2M.err
echo "cluster rc=$? after $(el): $(cut -c1-900 $OUT/bench_cluster_2M.json)"
grep -E "vsx_cluster_fast:" $OUT/bench_cluster_2M.err | tail -1 | cut -c1-400 | tee $OUT/bench_cluster_2M_phases.txt
timeout 900 python bench_allpairs.py --n 50000 --block 1000 --stream 1 > $OUT/bench_allpairs_50k.json 2> $OUT/bench_allpairs_50k.err
echo "allpairs 50k rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/bench_allpairs_50k.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:4], d['parity'])" 2>&1 | cut -c1-600)"
bash profiles/shim_cost.sh > $OUT/shim_cost.txt 2>&1; tail -12 $OUT/shim_cost.txt
echo "all done after $(el)"
