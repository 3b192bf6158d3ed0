#!/bin/bash
# profiles/run_r06m.sh -- r06: the ONE (single-strip) variants of the TILT kernels: whole -m gpu suite, aligner soak, default bench line
set -u
TAG=r06m
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests after $(el): $(tail -1 $OUT/tests.log)"; grep -E "FAILED|Error" $OUT/tests.log | head -5
bash profiles/run_soaks.sh $TAG/soaks 45 606 2>&1 | cut -c1-330; echo "soaks after $(el)"
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(el)"; python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"), d["end_to_end"].get("ms_calls"), "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
s = d.get("search_end_to_end", {})
print({k: s.get(k) for k in ("error", "queries_per_s", "queries_per_s_best", "seconds_later_calls")}, (s.get("reference_cli") or {}).get("same_hits_as_vsx"))
for k, v in d.get("shapes", {}).items(): print(k, v.get("value"), v.get("kernel_split_ms_per_step"), v.get("parity_all_fields_match"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("parity_all_fields_match"))
PY
