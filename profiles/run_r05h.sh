#!/bin/bash
# profiles/run_r05h.sh -- round 5: the search16 shim with two batches in flight (a context each): its tests, its soak, its cost.
set -u
TAG=r05h
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_shim.py -x -q > $OUT/tests.log 2>&1
echo "shim tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head
timeout 120 python oracle/soak_shim.py --seconds 60 --seed 20260927 --out gpurun_out/$TAG/soak_shim.json > $OUT/soak_shim.log 2>&1
echo "soak_shim rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/soak_shim.json')); print({k: v for k, v in d.items() if k not in ('failures','examples','what')})" 2>&1 | cut -c1-300)"
for rep in 1 2; do VSX_SHIM_STATS=1 GRAFT_REPO_ROOT=$REPO timeout 300 bash profiles/shim_cost.sh 2>&1 | cut -c1-300; done | tee $OUT/shim_cost.txt
echo "all done after $(el)"
