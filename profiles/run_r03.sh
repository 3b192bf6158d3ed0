#!/bin/bash
# profiles/run_r03.sh <tag> -- the round's evidence in one gpurun call: bench shape (bench line, rocprofv3 stats, PMC passes ->
# pmc_current.json), the config-3/4/5 shapes, the k-mer kernels, and kernel traces of the secondary kernels (DUST, ranking, MSA,
# tagged index build) through the commands that use them.  Everything lands under gpurun_out/<tag>*/.
set -u
TAG=${1:-r03q}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
FULL_BENCH=1 bash profiles/run_profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
bash profiles/run_shapes.sh ${TAG}_shapes 150x300x400000 300x300x400000 400x400x300000 > gpurun_out/${TAG}_shapes.log 2>&1
bash profiles/run_profile_kmer.sh ${TAG}_kmer > gpurun_out/${TAG}_kmer.log 2>&1
mkdir -p gpurun_out/${TAG}_misc
cd /tmp && export TMPDIR=/tmp
trace() {   # name, command...
  local name=$1; shift
  rm -rf /tmp/vsxmisc_$name
  rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/vsxmisc_$name -o t -- "$@" > $REPO/gpurun_out/${TAG}_misc/$name.log 2>&1
  for f in $(find /tmp/vsxmisc_$name -name "*kernel_stats.csv"); do grep -E "Name|vsx_" $f > $REPO/gpurun_out/${TAG}_misc/${name}_kernel_stats.csv; done
}
trace allpairs python $REPO/bench_allpairs.py --n 6000 --parity-prefix 0
trace search_dust python $REPO/bench.py --no-cpu --search-mask dust --steps 1 --warmup 0 --e2e-calls 1
trace cluster python $REPO/bench_cluster.py --n 200000 --parity-prefix 0
trace msa python $REPO/bench_msa.py
trace kmer_w12 python -c "
import sys; sys.path.insert(0, '$REPO')
import random
from tests import common
from vsearch_amd import Aligner, SearchSession
rng = random.Random(1)
anc = [common.rnd_seq(rng, 300) for _ in range(2000)]
db = [common.mutate(rng, anc[i % 2000], 0.05) for i in range(100000)]
qs = [common.mutate(rng, db[rng.randrange(len(db))][:200], 0.03) for _ in range(20000)]
with Aligner() as al:
    ss = SearchSession(al, db, id=0.9, wordlength=12)
    c = ss.candidates_batch(qs, device=True)
    print(len(c), ss.kmer_stats)
"
cat $REPO/gpurun_out/${TAG}_misc/*_kernel_stats.csv | cut -c1-160
