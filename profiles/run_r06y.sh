#!/bin/bash
# profiles/run_r06y.sh -- r06, after the last source changes (record-form window hits; comments / A/B macros in vsx_device.hip -> new kernel-source sha):
# the -m gpu suite, the PMC passes of `bench.py --kernels-only` (-> pmc_current.json on the new sha), the default bench line, search / api / shim soaks
set -u
TAG=r06y
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd $REPO
T0=$(date +%s); el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"; grep -E "FAILED|Error|assert" $OUT/tests.log | head -10
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
grep -E "vsx_forward|vsx_traceback_tilt" $OUT/summary.txt | head -8 | cut -c1-300
echo "trace + pmc done after $(el)"
cd $REPO
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(el)"; python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"), d["end_to_end"].get("ms_calls"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_note"))
s = d.get("search_end_to_end", {})
print({k: s.get(k) for k in ("error", "queries_per_s", "queries_per_s_best", "seconds_later_calls")}, (s.get("reference_cli") or {}).get("same_hits_as_vsx"))
for k, v in d.get("shapes", {}).items(): print(k, v.get("value"), v.get("kernel_split_ms_per_step"), v.get("parity_all_fields_match"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("parity_all_fields_match"))
PY
bash profiles/ab_search.sh $TAG/ab "VSX_X=0" "VSX_X=1"
for s in soak_search soak_api soak_shim; do
  python oracle/$s.py --seconds 40 --seed 6106 --out gpurun_out/$TAG/$s.json > gpurun_out/$TAG/$s.log 2>&1
  echo "$s rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/$s.json')); print({k: v for k, v in d.items() if k not in ('failures', 'examples', 'what')})" 2>&1 | cut -c1-300)"
done
echo "all done after $(el)"
