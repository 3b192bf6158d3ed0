#!/bin/bash
# profiles/run_r05o.sh -- round 5: lazy first batches whose second batch completes the reference's first eight (pairs aligned = a subset of the reference's):
# the search / shim / multirank tests, soak_search + soak_api, the default bench line
set -u
TAG=r05o
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_multirank.py tests/test_gpu_filters.py tests/test_gpu_mask.py tests/test_gpu_scale.py -x -q > $OUT/tests.log 2>&1
echo "search-side tests rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head
for s in soak_search soak_api; do
  timeout 120 python oracle/$s.py --seconds 40 --seed 20261001 --out gpurun_out/$TAG/$s.json > $OUT/$s.log 2>&1
  echo "$s rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/$s.json')); print({k: v for k, v in d.items() if k not in ('failures', 'examples', 'what')})" 2>&1 | cut -c1-300)"
done
python bench.py --no-shapes > $OUT/bench.json 2> $OUT/bench.err
echo "bench after $(( $(date +%s) - T0 )) s: $(python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); s=d['search_end_to_end']; print(d['value'], 'search', s['queries_per_s'], s['seconds_later_calls'], 'pairs', s['pairs_aligned'], 'hits', s['hits'], (s.get('reference_cli') or {}).get('same_hits_as_vsx'))" 2>&1 | cut -c1-400)"
