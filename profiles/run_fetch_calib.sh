#!/bin/bash
# profiles/run_fetch_calib.sh <tag> -- FETCH_SIZE calibration on the traceback's access patterns (vsearch_amd/csrc/ubench_fetch.hip):
# for every mode the timed run (lines/s) and a rocprofv3 --pmc FETCH_SIZE pass; prints counter bytes per touched line.
set -u
TAG=${1:-r04_fetch}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
BIN=$REPO/vsearch_amd/csrc/ubench_fetch
cd /tmp && export TMPDIR=/tmp
for M in 0 1 2 3 4; do
  $BIN $M 12 > $OUT/mode$M.txt 2>&1
  rm -rf /tmp/fc_$M
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d /tmp/fc_$M -o pmc -- $BIN $M 12 > $OUT/mode${M}_pmc.log 2>&1
  python - $M $OUT/mode$M.txt $(find /tmp/fc_$M -name "*counter_collection.csv" | head -1) <<'PY' | tee -a $OUT/summary.txt
import csv, re, sys
mode, txt, path = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else None)
line = open(txt).read().strip().splitlines()[-1]
lines = float(re.search(r"lines (\d+)", line).group(1))
print(line)
if path:
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r.get("Counter_Name") == "FETCH_SIZE" and ("sparse" in r.get("Kernel_Name", "") or "stream16" in r.get("Kernel_Name", ""))]
    for v in vals:
        print(f"  mode {mode}: FETCH_SIZE {v:.0f} KiB per dispatch = {v * 1024 / lines:.1f} counter bytes per touched 128-B line")
PY
done
