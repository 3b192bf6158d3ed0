#!/bin/bash
# profiles/run_r05l.sh -- round 5: lazy first batches in vsx_search_batch (a query's first batch = as many candidates as it still needs accepts):
# the whole -m gpu suite, soak_search + soak_api on a fresh seed, the default bench line with VSX_SEARCH_LAZY=0 and =1 on one box.
set -u
TAG=r05l
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head
for s in soak_search soak_api; do
  timeout 120 python oracle/$s.py --seconds 45 --seed 20260929 --out gpurun_out/$TAG/$s.json > $OUT/$s.log 2>&1
  echo "$s rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/$s.json')); print({k: v for k, v in d.items() if k not in ('failures', 'examples', 'what')})" 2>&1 | cut -c1-300)"
done
for LZ in 0 1 0 1; do
  VSX_SEARCH_LAZY=$LZ python bench.py --no-shapes > $OUT/bench_lazy$LZ.json 2> $OUT/bench_lazy$LZ.err
  echo "lazy=$LZ after $(el): $(python -c "import json; d=json.loads(open('$OUT/bench_lazy$LZ.json').read().strip().splitlines()[-1]); s=d['search_end_to_end']; print(d['value'], 'search', s['queries_per_s'], s['seconds_later_calls'], 'pairs', s['pairs_aligned'], 'hits', s['hits'], 'kmer', s['seconds_kmer'], 'align', s['seconds_align'], (s.get('reference_cli') or {}).get('same_hits_as_vsx'))" 2>&1 | cut -c1-400)"
done
echo "all done after $(el)"
