#!/bin/bash
# profiles/ab_tb.sh <tag> -- same-box A/B of the TILT-class traceback kernels (VSX_TB_V1=1: the first kernel) on the four BASELINE shapes,
# kernels only; prints the kernel split of every run
set -u
TAG=${1:-r04_ab}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for S in "250 1000 1000000" "150 300 400000" "300 300 400000" "400 400 300000"; do
  set -- $S
  for V in v1 v2; do
    if [ $V = v1 ]; then export VSX_TB_V1=1; else unset VSX_TB_V1; fi
    python bench.py --qlen $1 --dlen $2 --db $3 --kernels-only --steps 5 --warmup 2 > $OUT/${1}x${2}_$V.json 2> $OUT/${1}x${2}_$V.err
    python - $OUT/${1}x${2}_$V.json $V <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["config"]["workload"][:48], "| value", d["value"], "| ms/step", d["ms_per_step"], "| split", d["kernel_split_ms_per_step"], "| parity", d.get("parity_all_fields_match"))
except Exception as e:
    print(sys.argv[2], "unreadable:", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
