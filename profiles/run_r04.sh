#!/bin/bash
# profiles/run_r04.sh <tag> -- the round's evidence in one gpurun call: the default bench line as the driver runs it, rocprofv3 kernel
# stats + PMC passes of the bench shape (-> pmc_current.json), and the config-3/4/5 shapes (bench line, kernel stats, SQ / FETCH / WRITE).
# Everything lands under gpurun_out/<tag>*/ ; copy what should be judged into profiles/r04/.
set -u
TAG=${1:-r04z}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
FULL_BENCH=1 bash profiles/run_profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
bash profiles/run_shapes.sh ${TAG}_shapes 150x300x400000 300x300x400000 400x400x300000 > gpurun_out/${TAG}_shapes.log 2>&1
grep -E "SHAPE|vsx_traceback|vsx_forward" gpurun_out/${TAG}_shapes.log | cut -c1-260
tail -5 gpurun_out/$TAG/summary.txt | cut -c1-400
