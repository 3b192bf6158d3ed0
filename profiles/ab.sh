#!/bin/bash
# profiles/ab.sh <tag> "<variant> ..." "<QxDxDB> ..." -- same-box A/B of library builds (make VARIANT=x EXTRA=...): kernel split per
# shape and build.  variant "base" = vsearch_amd/libvsx.so.  Writes gpurun_out/<tag>/ab.txt
TAG=$1; VARS=$2; SHAPES=${3:-"150x300x400000 300x300x400000 400x400x300000 250x1000x1000000"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/$TAG
OUT=$REPO/gpurun_out/$TAG/ab.txt
: > $OUT
for S in $SHAPES; do
  Q=${S%%x*}; REST=${S#*x}; D=${REST%%x*}; DB=${REST#*x}
  for V in $VARS; do
    LIB=$REPO/vsearch_amd/libvsx_$V.so; [ "$V" = "base" ] && LIB=$REPO/vsearch_amd/libvsx.so
    VSX_LIBRARY=$LIB python $REPO/bench.py --qlen $Q --dlen $D --db $DB --kernels-only --steps ${STEPS:-5} --warmup 2 2> $REPO/gpurun_out/$TAG/err_${V}_${Q}x${D}.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_split_ms_per_step']
print('%-10s %-9s value %8.1f  ms/step %7.3f  fwd %7.3f  tb %7.3f  rest %6.3f' % ('$V','${Q}x${D}',d['value'],d['ms_per_step'],k['forward'],k['traceback'],k['cigar_text_and_rest']))
" >> $OUT 2>&1
  done
done
cat $OUT
