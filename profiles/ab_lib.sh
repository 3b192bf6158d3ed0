#!/bin/bash
# profiles/ab_lib.sh <tag> <variant...> -- same-box A/B of library builds (vsearch_amd/libvsx_<variant>.so; "default" = libvsx.so) on the
# bench shapes, kernels only
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
SHAPES=${SHAPES:-250x1000x1000000 150x300x400000 400x400x300000}
for S in $SHAPES; do
  Q=${S%%x*}; REST=${S#*x}; D=${REST%%x*}; DB=${REST#*x}
  for V in "$@"; do
    if [ $V = default ]; then unset VSX_LIBRARY; else export VSX_LIBRARY=$REPO/vsearch_amd/libvsx_$V.so; fi
    python bench.py --qlen $Q --dlen $D --db $DB --kernels-only --steps 5 --warmup 2 > $OUT/${Q}x${D}_$V.json 2> $OUT/${Q}x${D}_$V.err
    python - $OUT/${Q}x${D}_$V.json $V <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:8s}", d["config"]["workload"][34:52], "| value", d["value"], "| ms/step", d["ms_per_step"], "| split", d["kernel_split_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "unreadable:", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
