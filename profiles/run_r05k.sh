#!/bin/bash
# profiles/run_r05k.sh -- round 5: the pair-profile classes ON by default: the whole -m gpu suite, the aligner / allpairs / shim soaks on a fresh seed,
# allpairs at 20 000 sequences with the class off and on, the full 50 000 x 400 bp run.  Under gpurun_out/r05k/.
set -u
TAG=r05k
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head
for s in soak soak_allpairs soak_shim soak_cluster; do
  timeout 120 python oracle/$s.py --seconds 40 --seed 20260928 --out gpurun_out/$TAG/$s.json > $OUT/$s.log 2>&1
  echo "$s rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/$s.json')); print({k: v for k, v in d.items() if k not in ('failures', 'examples', 'what', 'scoring_kinds', 'shapes', 'by_command')})" 2>&1 | cut -c1-300)"
done
echo "soaks done after $(el)"
for PP in 0 1; do
  VSX_PAIRPROF=$PP timeout 600 python bench_allpairs.py --n 20000 --block 1000 --stream 1 --parity-prefix 0 > $OUT/allpairs_20k_pp$PP.json 2> $OUT/allpairs_20k_pp$PP.err
  echo "allpairs 20k pairprof=$PP rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/allpairs_20k_pp$PP.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['accepted_hits'], d['block_s'][:8])" 2>&1 | cut -c1-400)"
  tail -2 $OUT/allpairs_20k_pp$PP.err | cut -c1-300
done
timeout 900 python bench_allpairs.py --n 50000 --block 1000 --stream 1 > $OUT/bench_allpairs_50k.json 2> $OUT/bench_allpairs_50k.err
echo "allpairs 50k rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/bench_allpairs_50k.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:4], d['parity'])" 2>&1 | cut -c1-600)"
echo "all done after $(el)"
