#!/bin/bash
# profiles/run_r05d.sh -- round 5, fourth GPU call: (1) the traceback with its row checkpoints staged straight into LDS (libvsx_ldsst.so,
# -DVSX_TB_LDSSTAGE=1) against the default on the five pair shapes, two runs each, + its parity tests and a short soak; (2) the
# sparse-task classes on the other row classes: cands 1 / 2 / 4 at 300 x 300, 400 x 400 and 150 x 1000; (3) allpairs 20 k as a smoke of the
# inert-filter fast path (parity prefix).  Everything under gpurun_out/r05d/.
set -u
TAG=r05d
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
SHAPES="250x1000x1000000 150x1000x1000000 300x300x400000 400x400x300000 150x300x400000" bash profiles/ab_lib.sh $TAG/ldsst default ldsst default ldsst 2>&1 | tee $OUT/ldsst_ab.txt
echo "ldsst A/B done after $(el)"
VSX_LIBRARY=$REPO/vsearch_amd/libvsx_ldsst.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or torture or multi_strip or reference_batch or family or sparse_task or boundaries" > $OUT/ldsst_tests.log 2>&1
echo "ldsst tests rc=$? after $(el): $(tail -1 $OUT/ldsst_tests.log)"
VSX_LIBRARY=$REPO/vsearch_amd/libvsx_ldsst.so timeout 100 python oracle/soak.py --seconds 40 --seed 6161 --out gpurun_out/$TAG/ldsst_soak.json > $OUT/ldsst_soak.log 2>&1
echo "ldsst soak rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/ldsst_soak.json')); print({k: v for k, v in d.items() if k in ('rounds','pairs','mismatches','seed')})" 2>&1 | cut -c1-200)"
for S in 300x300x400000 400x400x300000 150x1000x1000000; do
  Q=${S%%x*}; REST=${S#*x}; D=${REST%%x*}; DB=${REST#*x}
  for C in 1 2 4; do
    for SP in 0 1; do
      VSX_SPARSE=$SP timeout 300 python bench.py --qlen $Q --dlen $D --db $DB --cands $C --kernels-only --steps 5 --warmup 2 > $OUT/s${Q}x${D}_c${C}_sp$SP.json 2> $OUT/s${Q}x${D}_c${C}_sp$SP.err
      python - $OUT/s${Q}x${D}_c${C}_sp$SP.json $Q $D $C $SP <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["roofline"]["plan"]
    print(f"{sys.argv[2]} x {sys.argv[3]} cands {sys.argv[4]} sparse {sys.argv[5]} | value {d['value']} | split {d['kernel_split_ms_per_step']} | R {p['rows_dominant']} tasks {p['tasks']} sparse {p.get('tasks_sparse')} waves {p.get('waves')}")
except Exception as e:
    print(sys.argv[2:], "unreadable:", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
    done
  done
done
echo "sparse shapes done after $(el)"
timeout 600 python bench_allpairs.py --n 20000 --block 1000 > $OUT/allpairs_20k.json 2> $OUT/allpairs_20k.err
echo "allpairs 20k rc=$? after $(el): $(cut -c1-900 $OUT/allpairs_20k.json)"
echo "all done after $(el)"
