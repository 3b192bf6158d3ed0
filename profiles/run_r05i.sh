#!/bin/bash
# profiles/run_r05i.sh -- round 5: the -m gpu suite, the smoke entry and the default bench line on the round's last commit (the shim back on one lane, the memory-pressure test)
set -u
TAG=r05i
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
timeout 700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log)"
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(( $(date +%s) - T0 )) s: $(python -c "import json; d=json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1]); print(d['value'], d['kernel_split_ms_per_step'], d.get('value_end_to_end'), d['roofline']['frac'], d['roofline'].get('traffic'), d['search_end_to_end'].get('queries_per_s'), d['cpu_baseline'].get('parity_all_fields_match'))" 2>&1 | cut -c1-300)"
