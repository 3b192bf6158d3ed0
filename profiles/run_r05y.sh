#!/bin/bash
# profiles/run_r05y.sh -- round 5, the final records AGAIN after the pair-profile classes went in (kernel sources changed), in the order that matters if the box time runs out:
# (1) the whole -m gpu suite, (2) rocprofv3 kernel trace + FETCH / WRITE / two SQ passes of `bench.py --kernels-only` (-> pmc_current.json),
# (3) the default bench line as the driver runs it, (4) BASELINE configs 3 and 4 at full size -- bench_cluster.py 2 M (parity on the first
# 100 000 sequences) and bench_allpairs.py 50 000 (parity on 1 124 250 pairs), (5) config 5's per-GPU share at 5 M x 1 kbp (search + aligner,
# reference CLI on 2 048 queries compared on query+target+id+caln), (6) the other pair shapes: bench line, kernel trace, SQ / FETCH / WRITE
# passes each, (7) bench_kmer.py.  Everything lands under gpurun_out/r05z/.
set -u
TAG=r05y
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"; free -g | head -2; git -C $REPO log -1 --format=%H 2>/dev/null; } > $OUT/host.txt 2>&1
timeout 700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
RC=$?
echo "tests rc=$RC after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head -10
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH --kernels-only > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $WORK/pmc_sq -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $WORK/pmc_sq2 -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq2.log 2>&1
python $REPO/profiles/summarize.py $WORK > $OUT/summary.txt 2>&1
for f in $(find $WORK/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
cp $WORK/traffic.json $OUT/traffic.json 2>/dev/null && cp $WORK/traffic.json $REPO/profiles/pmc_current.json
grep -E "vsx_forward|vsx_traceback_tilt|cigar_text" $OUT/summary.txt | head -12 | cut -c1-330
echo "trace + pmc done after $(el)"
cd $REPO
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(el)"; python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_note"))
s = d.get("search_end_to_end", {})
print({k: s.get(k) for k in ("error", "queries_per_s", "seconds_later_calls")}, (s.get("reference_cli") or {}).get("same_hits_as_vsx"))
for k, v in d.get("shapes", {}).items(): print(k, v.get("value"), v.get("kernel_split_ms_per_step"), v.get("parity_all_fields_match"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("parity_all_fields_match"))
PY
VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 100000 > $OUT/bench_cluster_2M.json 2> $OUT/bench_cluster_2M.err
echo "cluster rc=$? after $(el): $(cut -c1-900 $OUT/bench_cluster_2M.json)"
grep -E "vsx_cluster_fast:" $OUT/bench_cluster_2M.err | tail -1 | cut -c1-400 | tee $OUT/bench_cluster_2M_phases.txt
timeout 900 python bench_allpairs.py --n 50000 --block 1000 --stream 1 > $OUT/bench_allpairs_50k.json 2> $OUT/bench_allpairs_50k.err
echo "allpairs 50k rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/bench_allpairs_50k.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:4], d['parity'])" 2>&1 | cut -c1-600)"
PMC=1 bash profiles/run_shapes.sh $TAG/shapes 150x1000x1000000 300x300x400000 400x400x300000 2>&1 | cut -c1-330 | tee $OUT/shapes.txt
echo "shapes done after $(el)"
echo "all done after $(el)"

