#!/bin/bash
# profiles/run_r04w.sh -- the last 100 s of the round's GPU time, after the uniform-step-counter change of the DP kernel: the PMC passes
# that key profiles/pmc_current.json to the sources (FETCH_SIZE, WRITE_SIZE, the first SQ set), then as many of the search / filter
# tests as fit (the aligner's parity tests ran on this build in profiles/ab_uni.sh's call).
set -u
TAG=r04w
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --kernels-only --steps 1 --warmup 0"
timeout 25 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 25 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 25 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $WORK/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
python $REPO/profiles/summarize.py $WORK > $OUT/summary.txt 2>&1
cp $WORK/traffic.json $OUT/ 2>/dev/null
grep -E "vsx_forward|vsx_traceback_tilt" $OUT/summary.txt | cut -c1-260
echo "passes done after $(( $(date +%s) - T0 )) s"
cd $REPO
timeout 42 python -m pytest tests/test_gpu_search.py tests/test_gpu_filters.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$? (124 = cut by the clock): $(tail -2 $OUT/tests.log | tr '\n' ' ' | cut -c1-300)"
echo "done after $(( $(date +%s) - T0 )) s"
