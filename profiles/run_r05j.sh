#!/bin/bash
# profiles/run_r05j.sh -- round 5: the pair-profile classes (vsx_forward_kernel PAIR, VSX_PAIRPROF=1) on the GPU for the first time: parity test,
# same-box A/B on three pair shapes with 32 candidates per query (kernels only), allpairs at 20 000 sequences.  Under gpurun_out/r05j/.
set -u
TAG=r05j
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pair_profile" > $OUT/tests.log 2>&1
echo "pair tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert|INFO" $OUT/tests.log | head -10
for S in 400x400x300000 300x300x400000 250x1000x1000000 150x1000x1000000; do
  Q=${S%%x*}; REST=${S#*x}; D=${REST%%x*}; DB=${REST#*x}
  for PP in 0 1 0 1; do
    VSX_PAIRPROF=$PP timeout 300 python bench.py --qlen $Q --dlen $D --db $DB --queries 25000 --cands 32 --kernels-only --steps 5 --warmup 2 > $OUT/s${Q}x${D}_pp$PP.json 2> $OUT/s${Q}x${D}_pp$PP.err
    python - $OUT/s${Q}x${D}_pp$PP.json $Q $D $PP <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["roofline"]["plan"]
    print(f"{sys.argv[2]} x {sys.argv[3]} pairprof {sys.argv[4]} | value {d['value']} | split {d['kernel_split_ms_per_step']} | R {p['rows_dominant']} tasks {p['tasks']} pair {p.get('tasks_pair')}")
except Exception as e:
    print(sys.argv[2:], "unreadable:", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
  done
done
echo "A/B done after $(el)"
for PP in 0 1; do
  VSX_PAIRPROF=$PP timeout 600 python bench_allpairs.py --n 20000 --block 1000 --stream 1 --parity-prefix $([ $PP = 1 ] && echo 1500 || echo 0) > $OUT/allpairs_20k_pp$PP.json 2> $OUT/allpairs_20k_pp$PP.err
  echo "allpairs 20k pairprof=$PP rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/allpairs_20k_pp$PP.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['accepted_hits'], d['block_s'][:8], d['parity'])" 2>&1 | cut -c1-500)"
done
echo "all done after $(el)"
