#!/bin/bash
# profiles/run_r05p.sh -- round 5, the last call: the -m gpu suite, the smoke entry and the default bench line on the round's last commit;
# config 5's per-GPU share once more with lazy first batches in the search (no reference CLI leg this time: its 4 minutes of index build are in r05z)
set -u
TAG=r05p
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
echo "bench rc=$? after $(( $(date +%s) - T0 )) s: $(python -c "import json; d=json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1]); s=d['search_end_to_end']; print(d['value'], d['kernel_split_ms_per_step'], 'e2e', d.get('value_end_to_end'), 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), 'search', s['queries_per_s'], s['pairs_aligned'], (s.get('reference_cli') or {}).get('same_hits_as_vsx'), {k: (v['value'], v.get('parity_all_fields_match')) for k, v in d['shapes'].items()})" 2>&1 | cut -c1-700)"
VSX_BENCH_SEARCH_REPS=3 timeout 600 python bench.py --queries 1250000 --qlen 150 --db 5000000 --dlen 1000 --steps 2 --warmup 1 \
    --no-shapes --ref-search-queries 0 --e2e-calls 1 > $OUT/config5_share_lazy.json 2> $OUT/config5_share_lazy.err
echo "config5 rc=$? after $(( $(date +%s) - T0 )) s: $(python -c "import json; d=json.loads(open('$OUT/config5_share_lazy.json').read().strip().splitlines()[-1]); s=d['search_end_to_end']; print(d['value'], 'search', s.get('queries_per_s'), s.get('seconds'), s.get('pairs_aligned'), s.get('hits'), s.get('seconds_kmer'), s.get('seconds_align'), s.get('error'))" 2>&1 | cut -c1-400)"
