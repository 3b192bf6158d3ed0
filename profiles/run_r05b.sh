#!/bin/bash
# profiles/run_r05b.sh -- round 5, second GPU call: the sparse-task classes (vsx_forward_kernel NQ) and the feed record (VSX_FEED2) on
# the GPU for the first time.  (1) the whole -m gpu suite on the default build, (2) bench.py --cands 1 / 2 / 4 / 5 / 8 with and without
# the sparse classes (kernels only, same box), (3) same-box A/B of the feed2 build on the four pair shapes + its parity tests,
# (4) the default bench line (with the traceback of the search leg that failed in r05a), (5) config 5's per-GPU share again (memory
# pressure hook), (6) --cluster_fast at 2 M sequences: sparse classes off / on / on + speculative overlap.  Everything under gpurun_out/r05b/.
set -u
TAG=r05b
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head -20
# ---- (2) sparse tasks
for C in 1 2 4 5 8; do
  for SP in 0 1; do
    if [ $C = 8 ] && [ $SP = 0 ]; then continue; fi
    VSX_SPARSE=$SP timeout 300 python bench.py --cands $C --kernels-only --steps 5 --warmup 2 > $OUT/cands${C}_sparse$SP.json 2> $OUT/cands${C}_sparse$SP.err
    python - $OUT/cands${C}_sparse$SP.json $C $SP <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["roofline"]["plan"]
    print(f"cands {sys.argv[2]} sparse {sys.argv[3]} | value {d['value']} | ms/step {d['ms_per_step']} | split {d['kernel_split_ms_per_step']} | tasks {p['tasks']} sparse {p.get('tasks_sparse')} waves {p.get('waves')}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "unreadable:", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
echo "sparse A/B done after $(el)"
# ---- (3) feed2
SHAPES="250x1000x1000000 150x1000x1000000 300x300x400000 400x400x300000 150x300x400000" bash profiles/ab_lib.sh $TAG/feed2 default feed2 default feed2 2>&1 | tee $OUT/feed2_ab.txt
VSX_LIBRARY=$REPO/vsearch_amd/libvsx_feed2.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or torture or multi_strip or reference_batch or family or sparse_task or boundaries" > $OUT/feed2_tests.log 2>&1
echo "feed2 tests rc=$? after $(el): $(tail -1 $OUT/feed2_tests.log)"
VSX_LIBRARY=$REPO/vsearch_amd/libvsx_feed2.so timeout 100 python oracle/soak.py --seconds 40 --seed 5151 --out gpurun_out/$TAG/feed2_soak.json > $OUT/feed2_soak.log 2>&1
echo "feed2 soak rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/feed2_soak.json')); print({k: v for k, v in d.items() if k in ('rounds','pairs','mismatches','seed')})" 2>&1 | cut -c1-200)"
# ---- (4) default bench line
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(el)"; python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"))
s = d.get("search_end_to_end", {})
print({k: s.get(k) for k in ("error", "traceback", "queries_per_s", "seconds_later_calls", "reference_cli")})
PY
# ---- (5) config 5 share
VSX_BENCH_SEARCH_REPS=3 timeout 1500 python bench.py --queries 1250000 --qlen 150 --db 5000000 --dlen 1000 --steps 2 --warmup 1 \
    --no-shapes --ref-search-queries 2048 --e2e-calls 1 > $OUT/config5_share.json 2> $OUT/config5_share.err
echo "config5 rc=$? after $(el)"; python - $OUT/config5_share.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"))
print(json.dumps(d.get("search_end_to_end"))[:3000])
PY
tail -3 $OUT/config5_share.err
# ---- (6) cluster_fast 2 M
for V in "0 0" "1 0" "1 1"; do
  set -- $V
  VSX_SPARSE=$1 VSX_CLUSTER_SPEC_OVERLAP=$2 VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix $([ "$V" = "1 1" ] && echo 100000 || echo 0) \
      > $OUT/cluster_sparse$1_spec$2.json 2> $OUT/cluster_sparse$1_spec$2.err
  echo "cluster sparse=$1 spec=$2 rc=$? after $(el): $(cut -c1-700 $OUT/cluster_sparse$1_spec$2.json)"
  grep -E "cluster_fast:|phases|total" $OUT/cluster_sparse$1_spec$2.err | tail -4 | cut -c1-400
done
echo "all done after $(el)"
