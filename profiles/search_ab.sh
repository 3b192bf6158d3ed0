#!/bin/bash
# A/B of the search pipeline's knobs on the bench's --usearch_global leg (same box, one process per setting).
# usage (GPU box): bash profiles/search_ab.sh > gpurun_out/search_ab.txt
export VSX_BENCH_SEARCH_REPS=7
run() {
  echo "== $*"
  env "$@" python bench.py --steps 1 --warmup 0 --no-cpu --e2e-calls 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['search_end_to_end']
print(d['queries_per_s'], d['seconds'], sorted(d['seconds_later_calls']), d['hits'])"
}
run VSX_KMER_TURNS=0 VSX_SEARCH_TAPER=2
run VSX_KMER_TURNS=1 VSX_SEARCH_TAPER=2
run VSX_KMER_TURNS=1 VSX_SEARCH_TAPER=3
run VSX_KMER_TURNS=1 VSX_SEARCH_TAPER=3 VSX_SEARCH_RANKERS=3
run VSX_KMER_TURNS=0 VSX_SEARCH_TAPER=3
run VSX_KMER_TURNS=1 VSX_SEARCH_TAPER=2
