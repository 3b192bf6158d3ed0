#!/bin/bash
# profiles/run_r05e.sh -- round 5, fifth GPU call: (1) where the bytes of global_load_lds land (ubench_ldsdma), (2) the -m gpu suite on the
# planner's sparse-class threshold + merged traceback launches, (3) --cluster_fast 2 M sparse off / on, (4) the per-slice timeline of the
# first two blocks of --allpairs_global 50 000 x 400 bp.  Everything under gpurun_out/r05e/.
set -u
TAG=r05e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
timeout 60 vsearch_amd/csrc/ubench_ldsdma > $OUT/ubench_ldsdma.txt 2>&1; cat $OUT/ubench_ldsdma.txt
timeout 700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"
grep -E "FAILED|Error|assert" $OUT/tests.log | head -20
for SP in 0 1; do
  VSX_SPARSE=$SP VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 0 > $OUT/cluster_sparse$SP.json 2> $OUT/cluster_sparse$SP.err
  echo "cluster sparse=$SP rc=$? after $(el): $(cut -c1-500 $OUT/cluster_sparse$SP.json)"
  grep -E "vsx_cluster_fast:" $OUT/cluster_sparse$SP.err | tail -1 | cut -c1-400
done
VSX_DEBUG_TIMING=1 timeout 600 python bench_allpairs.py --n 50000 --block 1000 --max-blocks 2 --parity-prefix 0 > $OUT/allpairs_2blocks.json 2> $OUT/allpairs_2blocks.err
echo "allpairs 2 blocks rc=$? after $(el): $(cut -c1-600 $OUT/allpairs_2blocks.json)"
grep -E "slice|vsx_align_pairs" $OUT/allpairs_2blocks.err | head -150 > $OUT/allpairs_timeline.txt
tail -30 $OUT/allpairs_timeline.txt
echo "all done after $(el)"
