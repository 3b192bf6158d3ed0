#!/bin/bash
# profiles/run_r05c.sh -- round 5, third GPU call: the sparse-task classes rebuilt on wave-shared checkpoint blocks (full-line stores), FEED2 on
# by default, the search16 combiner of the shim, the device-memory reserve.  (1) -m gpu suite, (2) bench.py --cands 1 / 2 / 4 / 5 with and
# without the sparse classes, (3) the relinked CLI's cost (profiles/shim_cost.sh), (4) the default bench line, (5) config 5's per-GPU share,
# (6) --cluster_fast 2 M: sparse off / on / on + speculative overlap.  Everything under gpurun_out/r05c/.
set -u
TAG=r05c
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head -20
for C in 1 2 4 5; do
  for SP in 0 1; do
    VSX_SPARSE=$SP timeout 300 python bench.py --cands $C --kernels-only --steps 5 --warmup 2 > $OUT/cands${C}_sparse$SP.json 2> $OUT/cands${C}_sparse$SP.err
    python - $OUT/cands${C}_sparse$SP.json $C $SP <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["roofline"]["plan"]
    print(f"cands {sys.argv[2]} sparse {sys.argv[3]} | value {d['value']} | ms/step {d['ms_per_step']} | split {d['kernel_split_ms_per_step']} | tasks {p['tasks']} sparse {p.get('tasks_sparse')} waves {p.get('waves')} | ckpt GB {d['roofline']['hbm_algorithmic_bytes_per_launch'] / 1e9:.1f}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "unreadable:", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
echo "sparse A/B done after $(el)"
VSX_SHIM_STATS=1 GRAFT_REPO_ROOT=$REPO timeout 300 bash profiles/shim_cost.sh > $OUT/shim_cost.txt 2>&1
cat $OUT/shim_cost.txt | cut -c1-300
echo "shim cost done after $(el)"
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(el)"; python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"))
s = d.get("search_end_to_end", {})
print({k: s.get(k) for k in ("error", "traceback", "queries_per_s", "seconds_later_calls", "reference_cli")})
for k, v in d.get("shapes", {}).items(): print(k, v.get("value"), v.get("kernel_split_ms_per_step"), v.get("parity_all_fields_match"))
PY
VSX_BENCH_SEARCH_REPS=3 timeout 1500 python bench.py --queries 1250000 --qlen 150 --db 5000000 --dlen 1000 --steps 2 --warmup 1 \
    --no-shapes --ref-search-queries 2048 --e2e-calls 1 > $OUT/config5_share.json 2> $OUT/config5_share.err
echo "config5 rc=$? after $(el)"; python - $OUT/config5_share.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"))
    print(json.dumps(d.get("search_end_to_end"))[:3000])
except Exception as e:
    print("config5 unreadable", e)
PY
tail -3 $OUT/config5_share.err | cut -c1-400
for V in "0 0" "1 0" "1 1"; do
  set -- $V
  VSX_SPARSE=$1 VSX_CLUSTER_SPEC_OVERLAP=$2 VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix $([ "$V" = "1 0" ] && echo 100000 || echo 0) \
      > $OUT/cluster_sparse$1_spec$2.json 2> $OUT/cluster_sparse$1_spec$2.err
  echo "cluster sparse=$1 spec=$2 rc=$? after $(el): $(cut -c1-700 $OUT/cluster_sparse$1_spec$2.json)"
  grep -E "vsx_cluster_fast:" $OUT/cluster_sparse$1_spec$2.err | tail -1 | cut -c1-400
done
echo "all done after $(el)"
