#!/bin/bash
# profiles/run_r04y.sh -- the round's last GPU call: the whole GPU suite on the DP kernel with the shorter last-row tracking, then the
# evidence in the order that matters if the box time runs out: kernel trace + FETCH / WRITE passes (-> pmc_current.json), the default
# bench line as the driver runs it, the two SQ passes, a short fresh-seed aligner soak.  Everything lands under gpurun_out/r04y/.
set -u
TAG=r04y
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd $REPO
T0=$(date +%s)
timeout 330 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
RC=$?
echo "tests rc=$RC after $(( $(date +%s) - T0 )) s: $(tail -1 $OUT/tests.log)"
if [ $RC -ne 0 ]; then tail -40 $OUT/tests.log; exit 1; fi
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH --kernels-only > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
python $REPO/profiles/summarize.py $WORK > $OUT/summary_early.txt 2>&1
cp $WORK/traffic.json $OUT/traffic_early.json 2>/dev/null && cp $WORK/traffic.json $REPO/profiles/pmc_current.json
echo "trace + fetch + write done after $(( $(date +%s) - T0 )) s"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
echo "bench done after $(( $(date +%s) - T0 )) s"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $WORK/pmc_sq -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $WORK/pmc_sq2 -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq2.log 2>&1
python $REPO/profiles/summarize.py $WORK > $OUT/summary.txt 2>&1
for f in $(find $WORK -name "*kernel_stats.csv"); do cp $f $OUT/; done
cp $WORK/traffic.json $OUT/ 2>/dev/null
grep -E "vsx_forward|vsx_traceback_tilt|cigar_text" $OUT/summary.txt | head -12 | cut -c1-330
echo "SQ passes done after $(( $(date +%s) - T0 )) s"
cd $REPO
timeout 70 python oracle/soak.py --seconds 40 --seed 4242 --out gpurun_out/$TAG/soak_aligner.json > $OUT/soak.log 2>&1
echo "soak rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/soak_aligner.json')); print({k: v for k, v in d.items() if k in ('rounds','pairs','mismatches','seed')})" 2>&1 | cut -c1-200)"
echo "all done after $(( $(date +%s) - T0 )) s"
