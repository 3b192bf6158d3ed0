#!/bin/bash
# profiles/run_soaks.sh <tag> <seconds per soak> <seed> -- every randomized soak against the reference, fresh seed; records -> gpurun_out/<tag>/
TAG=${1:-r03_soak}; SECS=${2:-60}; SEED=${3:-33}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
for s in soak soak_search soak_cluster soak_allpairs soak_api soak_shim; do
  python oracle/$s.py --seconds $SECS --seed $SEED --out gpurun_out/$TAG/$s.json > gpurun_out/$TAG/$s.log 2>&1
  echo "$s rc=$? $(python -c "import json; d=json.load(open('gpurun_out/$TAG/$s.json')); print({k: v for k, v in d.items() if k not in ('failures', 'examples', 'what')})" 2>&1 | cut -c1-300)"
done
