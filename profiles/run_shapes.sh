#!/bin/bash
# profiles/run_shapes.sh <tag> [QxDxDB ...] -- per-shape evidence on the GPU box (via gpurun): for every shape a bench line
# (kernels only), the rocprofv3 kernel-trace stats of the same command and two PMC passes (SQ set, FETCH_SIZE).
# Writes gpurun_out/<tag>/<shape>/{bench.json,summary.txt,*kernel_stats.csv}; copy what should be judged into profiles/.
set -u
TAG=${1:-r03}
shift
SHAPES=${@:-"150x300x400000 300x300x400000 400x400x300000 250x1000x1000000"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for S in $SHAPES; do
  Q=${S%%x*}; REST=${S#*x}; D=${REST%%x*}; DB=${REST#*x}
  OUT=$REPO/gpurun_out/$TAG/${Q}x${D}
  WORK=/tmp/vsxshape_${TAG}_${Q}x${D}
  rm -rf $WORK; mkdir -p $OUT $WORK
  BENCH="python $REPO/bench.py --qlen $Q --dlen $D --db $DB --kernels-only"
  $BENCH --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
  rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH --steps 3 --warmup 1 > $OUT/trace.log 2>&1
  if [ "${PMC:-1}" = "1" ]; then
    rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $WORK/pmc_sq -o pmc -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
    rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
    rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
  fi
  VSX_SUMMARY_WORKLOAD="$Q,$D,$DB" python $REPO/profiles/summarize.py $WORK 2>&1 | grep -v "at::native\|rocprim\|rocclr\|anonymous" > $OUT/summary.txt
  for f in $(find $WORK -name "*kernel_stats.csv"); do grep -E "^\"?Name|vsx_" $f > $OUT/kernel_stats.csv; done
  python - "$OUT/bench.json" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("SHAPE", d["config"]["workload"][:60], "| value", d["value"], "GCUPS | ms/step", d["ms_per_step"], "| split", d["kernel_split_ms_per_step"],
          "| kernel", r["kernel"], r["kernel_gcups"], "GCUPS frac", r["frac"])
except Exception as e:
    print("bench line unreadable:", e)
EOF
  grep -E "vsx_" $OUT/summary.txt | head -12
done
