#!/bin/bash
# profiles/kmer_probe.sh [lib variants...] -- k-mer count kernel under VSX_KMER_PROBE (1 = no LDS atomics, 2 = no postings loads, 3 = neither)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for V in ${@:-base}; do
  LIB=$REPO/vsearch_amd/libvsx_$V.so; [ "$V" = "base" ] && LIB=$REPO/vsearch_amd/libvsx.so
  for p in ${PROBES:-0 1 2 3}; do
    VSX_LIBRARY=$LIB VSX_KMER_PROBE=$p python $REPO/bench_kmer.py --host-queries 0 --repeat 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V probe $p', d['kernel_ms'], 'ms', d['roofline']['achieved'], 'GB/s')"
  done
done
