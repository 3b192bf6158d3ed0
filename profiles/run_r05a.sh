#!/bin/bash
# profiles/run_r05a.sh -- round 5, first GPU call (kernel sources = the r04 final state): the records VERDICT r04 "next 1" asks for
# that do not depend on this round's kernel work: (1) the default bench line with the config-5 pair shape as BASELINE states it
# (150 x 1000) and the reference CLI compared on query+target+id+caln, (2) rocprofv3 kernel trace + PMC passes of the same command,
# (3) BASELINE config 5's per-GPU share at full size: 1.25 M x 150 bp queries against 5 M x 1 kbp.  Everything under gpurun_out/r05a/.
set -u
TAG=r05a
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd $REPO
T0=$(date +%s)
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"; free -g | head -2; df -h /tmp | tail -1; rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -1; } > $OUT/host.txt 2>&1
cat $OUT/host.txt
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(( $(date +%s) - T0 )) s"; tail -c 1500 $OUT/bench_full.json
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH --kernels-only > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $WORK/pmc_sq -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
python $REPO/profiles/summarize.py $WORK > $OUT/summary.txt 2>&1
for f in $(find $WORK/trace -name "*kernel_stats.csv"); do cp $f $OUT/; done
cp $WORK/traffic.json $OUT/ 2>/dev/null
grep -E "vsx_forward|vsx_traceback_tilt|cigar_text" $OUT/summary.txt | head -8 | cut -c1-300
echo "trace + pmc done after $(( $(date +%s) - T0 )) s"
cd $REPO
# BASELINE config 5, one GPU's share of the 8-GPU job (10 M queries / 8), the database at its stated size
VSX_BENCH_SEARCH_REPS=3 timeout 1500 python bench.py --queries 1250000 --qlen 150 --db 5000000 --dlen 1000 --steps 2 --warmup 1 \
    --no-shapes --ref-search-queries 2048 --e2e-calls 1 > $OUT/config5_share.json 2> $OUT/config5_share.err
echo "config5 rc=$? after $(( $(date +%s) - T0 )) s"; tail -c 3000 $OUT/config5_share.json; tail -5 $OUT/config5_share.err
free -g | head -2 >> $OUT/host.txt
echo "all done after $(( $(date +%s) - T0 )) s"
