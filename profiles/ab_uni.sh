#!/bin/bash
# profiles/ab_uni.sh -- same-box A/B of the "uniform step counter" DP build (experiments/dp_uniform_steps.patch -> vsearch_amd/libvsx_uni.so)
# against libvsx.so on a partial-task workload (5 candidates per query: every task has an empty target group), on the bench shape, and
# the parity tests of the aligner on the variant.  Evidence for the next round; the variant is not the shipped library.
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
