#!/bin/bash
# profiles/ab_e2e.sh <tag> "<env assignments>" ... -- vsx_align_pairs end to end (bench.py's end_to_end leg) under different environments
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
n=0
for E in "$@"; do
  n=$((n + 1))
  env $E python bench.py --no-cpu --no-search --steps 2 --warmup 1 --e2e-calls 7 > $OUT/run$n.json 2> $OUT/run$n.err
  python - $OUT/run$n.json "$E" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e = d["end_to_end"]
    print(f"{sys.argv[2]:40s} | kernels {d['ms_per_step']} ms | e2e {e['value']} GCUPS, {e['ms_per_call']} ms per call (min {e['ms_per_call_min']}), equal {e['equals_plan_results']}")
except Exception as ex:
    print(sys.argv[2], "unreadable:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
