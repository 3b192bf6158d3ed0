#!/bin/bash
# profiles/run_profile.sh <tag> -- run on the GPU box (via gpurun): bench line + rocprofv3 kernel trace + PMC passes.
# Writes everything under gpurun_out/<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" > $OUT/host.txt
python -c "import os; print('affinity', len(os.sched_getaffinity(0)))" >> $OUT/host.txt; cat $OUT/host.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1"
if [ "${FULL_BENCH:-0}" = "1" ]; then $BENCH > $OUT/bench.json 2> $OUT/bench.err; else $BENCH --no-cpu --no-search > $OUT/bench.json 2> $OUT/bench.err; fi
tail -c 3000 $OUT/bench.json
# per-kernel time (same command, no CPU leg)
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH --kernels-only > $OUT/trace.log 2>&1
# counters, each in its own pass (TCC: FETCH_SIZE takes 3 slots, WRITE_SIZE 2)
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $WORK/pmc_sq -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $WORK/pmc_sq2 -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq2.log 2>&1
find $WORK -name "*.csv" | xargs ls -la | head -30
python $REPO/profiles/summarize.py $WORK > $OUT/summary.txt 2>&1
for f in $(find $WORK -name "*kernel_stats.csv"); do cp $f $OUT/; done
cat $OUT/summary.txt
cp $WORK/traffic.json $OUT/ 2>/dev/null
# the file bench.py reads (it refuses it once the kernel sources change): commit profiles/pmc_current.json with the kernels it was measured on
cp $WORK/traffic.json $REPO/gpurun_out/pmc_current.json 2>/dev/null
