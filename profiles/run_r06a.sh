#!/bin/bash
# profiles/run_r06a.sh -- round 6, first contact: (1) where a --usearch_global call spends its wall time (VSX_DEBUG_TIMELINE per window,
# VSX_DEBUG_TIMING totals, 7 calls), (2) rocprofv3 kernel trace + FETCH / SQ / LDS-conflict PMC of the k-mer kernels on HEAD (none since r03),
# (3) the default bench line on this box as the round's baseline.  Everything lands under gpurun_out/r06a/.
set -u
TAG=r06a
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"; free -g | head -2; } > $OUT/host.txt 2>&1
# (1) search timeline
VSX_DEBUG_TIMELINE=1 VSX_DEBUG_TIMING=1 VSX_BENCH_SEARCH_REPS=7 python bench.py --no-cpu --no-shapes --steps 1 --warmup 0 --e2e-calls 1 --ref-search-queries 0 \
    > $OUT/search_timeline.json 2> $OUT/search_timeline.err
echo "search timeline rc=$? after $(el)"
python - $OUT/search_timeline.json <<'PY'
import json, sys
s = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["search_end_to_end"]
print({k: s.get(k) for k in ("queries_per_s", "seconds", "seconds_median", "seconds_later_calls", "hits", "seconds_kmer_kernel")})
PY
grep -E "vsx_search_batch:" $OUT/search_timeline.err | tail -4
# (2) k-mer kernels: trace + PMC
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench_kmer.py --host-queries 0 --repeat 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH > $OUT/kmer_trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH > $OUT/kmer_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH > $OUT/kmer_pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $WORK/pmc_sq -o pmc -- $BENCH > $OUT/kmer_pmc_sq.log 2>&1
python $REPO/profiles/summarize.py $WORK 2>&1 | grep -E "^==|vsx_" > $OUT/kmer_summary.txt
for f in $(find $WORK/trace -name "*kernel_stats.csv"); do grep -E "Name|vsx_" $f > $OUT/kmer_kernel_stats.csv; done
cut -c1-300 $OUT/kmer_summary.txt | head -40
echo "kmer pmc done after $(el)"
cd $REPO
timeout 300 python bench_kmer.py > $OUT/bench_kmer.json 2> $OUT/bench_kmer.err
echo "kmer rc=$? after $(el): $(cut -c1-600 $OUT/bench_kmer.json)"
# (3) the default line
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(el)"; python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"), "frac", d["roofline"]["frac"])
s = d.get("search_end_to_end", {})
print({k: s.get(k) for k in ("error", "queries_per_s", "seconds_median", "seconds_later_calls")}, (s.get("reference_cli") or {}).get("same_hits_as_vsx"))
for k, v in d.get("shapes", {}).items(): print(k, v.get("value"), v.get("kernel_split_ms_per_step"), v.get("parity_all_fields_match"))
PY
echo "all done after $(el)"
