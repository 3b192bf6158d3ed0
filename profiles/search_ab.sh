#!/bin/bash
# A/B of search pipeline / k-mer index knobs on the bench's --usearch_global leg (same box, one process per setting).
# usage (GPU box): SETTINGS="VSX_KMER_PACKED=0;VSX_KMER_PACKED=1" bash profiles/search_ab.sh > gpurun_out/search_ab.txt
export VSX_BENCH_SEARCH_REPS=${REPS:-7}
run() {
  echo "== $*"
  env "$@" python bench.py --steps 1 --warmup 0 --no-cpu --e2e-calls 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['search_end_to_end']
print(d['queries_per_s'], d['seconds'], sorted(d['seconds_later_calls']), d['hits'], 'create', d['searcher_create_s'], 'first', d['seconds_first_call'])"
}
IFS=';' read -ra SET <<< "${SETTINGS:-VSX_KMER_TURNS=1}"
for s in "${SET[@]}"; do run $s; done
