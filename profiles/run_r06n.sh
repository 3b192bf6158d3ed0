#!/bin/bash
# profiles/run_r06n.sh -- r06 A/B: R >= 26 at three waves per SIMD (168 VGPRs) now that the ONE variants need fewer registers (r03: 3 waves spilled into the loop)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06n; mkdir -p $OUT; cd $REPO
run() {  # lib label, bench args
  LIB=$REPO/vsearch_amd/libvsx.so; [ "$1" != base ] && LIB=$REPO/vsearch_amd/libvsx_$1.so
  VSX_LIBRARY=$LIB python bench.py "${@:3}" --kernels-only --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_split_ms_per_step']
print('%-6s %-22s value %8.1f  fwd %7.3f  tb %7.3f' % ('$1','$2',d['value'],k['forward'],k['traceback']))"
}
for rep in 1 2; do
for V in base w3; do
  run $V 400x400 --qlen 400 --dlen 400 --db 300000
  run $V 400x400_dense32 --qlen 400 --dlen 400 --db 300000 --queries 25000 --cands 32
  run $V 500x500 --qlen 500 --dlen 500 --db 300000
  run $V 440x440 --qlen 440 --dlen 440 --db 300000
done; done
