#!/bin/bash
# profiles/run_r05f.sh -- round 5, sixth GPU call: (1) the traceback with LDS-staged row checkpoints, second build (16-byte lane stride of
# global_load_lds_dwordx3, one LDS arena) against the default: five pair shapes, two runs each, parity tests, aligner soak; (2) the
# allpairs stream: its test, a same-box A/B at 20 000 sequences (block calls / stream), the full 50 000 x 400 bp run.  Under gpurun_out/r05f/.
set -u
TAG=r05f
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
VSX_LIBRARY=$REPO/vsearch_amd/libvsx_ldsst.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or torture or multi_strip or reference_batch or family or sparse_task or boundaries" > $OUT/ldsst_tests.log 2>&1
echo "ldsst tests rc=$? after $(el): $(tail -1 $OUT/ldsst_tests.log)"
VSX_LIBRARY=$REPO/vsearch_amd/libvsx_ldsst.so timeout 100 python oracle/soak.py --seconds 40 --seed 6161 --out gpurun_out/$TAG/ldsst_soak.json > $OUT/ldsst_soak.log 2>&1
echo "ldsst soak rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/ldsst_soak.json')); print({k: v for k, v in d.items() if k in ('rounds','pairs','mismatches','seed')})" 2>&1 | cut -c1-200)"
SHAPES="250x1000x1000000 150x1000x1000000 300x300x400000 400x400x300000 150x300x400000" bash profiles/ab_lib.sh $TAG/ldsst default ldsst default ldsst 2>&1 | tee $OUT/ldsst_ab.txt
echo "ldsst A/B done after $(el)"
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_scale.py -x -q -k "allpairs" > $OUT/allpairs_tests.log 2>&1
echo "allpairs tests rc=$? after $(el): $(tail -1 $OUT/allpairs_tests.log)"
grep -E "FAILED|Error|assert" $OUT/allpairs_tests.log | head
for ST in 0 1; do
  timeout 600 python bench_allpairs.py --n 20000 --block 1000 --stream $ST --parity-prefix 0 > $OUT/allpairs_20k_stream$ST.json 2> $OUT/allpairs_20k_stream$ST.err
  echo "allpairs 20k stream=$ST rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/allpairs_20k_stream$ST.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:6])" 2>&1 | cut -c1-300)"
done
timeout 900 python bench_allpairs.py --n 50000 --block 1000 --stream 1 > $OUT/allpairs_50k_stream.json 2> $OUT/allpairs_50k_stream.err
echo "allpairs 50k rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/allpairs_50k_stream.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['parity'])" 2>&1 | cut -c1-600)"
echo "all done after $(el)"
