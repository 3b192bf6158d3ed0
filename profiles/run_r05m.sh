#!/bin/bash
# profiles/run_r05m.sh -- round 5, last call: --cluster_fast 2 M with lazy first batches off / on (VSX_CLUSTER_LAZY), then the default bench line
# of the round's last commit (lazy search on), twice.
set -u
TAG=r05m
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
for LZ in 0 1; do
  VSX_CLUSTER_LAZY=$LZ VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix $([ $LZ = 1 ] && echo 100000 || echo 0) > $OUT/cluster_lazy$LZ.json 2> $OUT/cluster_lazy$LZ.err
  echo "cluster lazy=$LZ rc=$? after $(el): $(cut -c1-800 $OUT/cluster_lazy$LZ.json)"
  grep -E "vsx_cluster_fast:" $OUT/cluster_lazy$LZ.err | tail -1 | cut -c1-400
done
for rep in 1 2; do
  python bench.py > $OUT/bench_full_$rep.json 2> $OUT/bench_full_$rep.err
  echo "bench $rep rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/bench_full_$rep.json').read().strip().splitlines()[-1]); s=d['search_end_to_end']; print(d['value'], d['kernel_split_ms_per_step'], 'e2e', d.get('value_end_to_end'), 'frac', d['roofline']['frac'], 'search', s['queries_per_s'], s['seconds_later_calls'], s['pairs_aligned'], (s.get('reference_cli') or {}).get('same_hits_as_vsx'), {k: v['value'] for k, v in d['shapes'].items()})" 2>&1 | cut -c1-600)"
done
echo "all done after $(el)"
