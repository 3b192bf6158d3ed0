#!/bin/bash
# profiles/run_profile_kmer.sh <tag> -- rocprofv3 kernel trace + PMC passes of the k-mer counting bench (GPU box, via gpurun)
set -u
TAG=${1:-r01e}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench_kmer.py --host-queries 0 --repeat 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH > $OUT/kmer_trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH > $OUT/kmer_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $WORK/pmc_sq -o pmc -- $BENCH > $OUT/kmer_pmc_sq.log 2>&1
python $REPO/profiles/summarize.py $WORK 2>&1 | grep -E "^==|vsx_" > $OUT/kmer_summary.txt
for f in $(find $WORK -name "*kernel_stats.csv"); do grep -E "Name|vsx_" $f > $OUT/kmer_kernel_stats.csv; done
cat $OUT/kmer_summary.txt
