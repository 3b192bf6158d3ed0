#!/bin/bash
# profiles/ab_uni.sh -- how profiles/r04/r04w_uniform_steps_ab.txt was measured: same-box A/B of a library whose DP kernel lets every lane
# run the steady loop (vsearch_amd/libvsx_uni.so: vsx_device.hip with that one condition changed, linked with the other objects of the
# default build) against libvsx.so of commit 04baad9, on a partial-task workload (5 candidates per query: every task has an empty
# target group) and on the bench shape, plus the aligner's parity tests on the variant.  The change is in the sources since 0254dd2.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04w_uni
mkdir -p $OUT
cd $REPO
for C in 5 8; do
  for V in default uni; do
    if [ $V = default ]; then unset VSX_LIBRARY; else export VSX_LIBRARY=$REPO/vsearch_amd/libvsx_uni.so; fi
    python bench.py --cands $C --kernels-only --steps 4 --warmup 1 > $OUT/c${C}_$V.json 2> $OUT/c${C}_$V.err
    python - $OUT/c${C}_$V.json $V $C <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"cands {sys.argv[3]} {sys.argv[2]:8s} value", d["value"], "| ms/step", d["ms_per_step"], "| split", d["kernel_split_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "unreadable:", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
export VSX_LIBRARY=$REPO/vsearch_amd/libvsx_uni.so
timeout 75 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/parity_uni.log 2>&1
echo "parity on the variant rc=$?: $(tail -1 $OUT/parity_uni.log)"
