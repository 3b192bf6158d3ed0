#!/bin/bash
# profiles/run_r06z.sh -- round 6, the records on the round's final kernel sources, in the order that matters if the box time runs out:
# (1) the whole -m gpu suite, (2) rocprofv3 kernel trace + FETCH / WRITE / two SQ passes of `bench.py --kernels-only` (-> pmc_current.json),
# (3) the k-mer kernels: trace + FETCH / WRITE / SQ+LDS passes of bench_kmer.py (-> pmc_kmer_current.json), bench_kmer.py with roofline.traffic,
# (4) the default bench line as the driver runs it, (5) BASELINE configs[2] and [3] at full size -- bench_cluster.py 2 M (parity on the first
# 250 000 sequences) and bench_allpairs.py 50 000 (parity on the pairs among the first 3 000 sequences), (6) configs[4]'s per-GPU share at
# 5 M x 1 kbp, (7) the other pair shapes: bench line, kernel trace, SQ / FETCH / WRITE passes each, (8) the six soaks on a fresh seed,
# (9) two ranks over RCCL on this one GPU (--dry-collectives; expected to be refused by RCCL: recorded either way).
set -u
TAG=${TAG:-r06z}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
WORK=/tmp/vsxprof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd $REPO
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"; free -g | head -2; } > $OUT/host.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$? after $(el): $(tail -1 $OUT/tests.log)"; grep -E "FAILED|Error|assert" $OUT/tests.log | head -10
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $WORK/trace -o trace -- $BENCH --kernels-only > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $WORK/pmc_fetch -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $WORK/pmc_write -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $WORK/pmc_sq -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $WORK/pmc_sq2 -o pmc -- $BENCH --kernels-only --steps 1 --warmup 0 > $OUT/pmc_sq2.log 2>&1
python $REPO/profiles/summarize.py $WORK > $OUT/summary.txt 2>&1
for f in $(find $WORK/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
cp $WORK/traffic.json $OUT/traffic.json 2>/dev/null && cp $WORK/traffic.json $REPO/profiles/pmc_current.json
grep -E "vsx_forward|vsx_traceback_tilt|cigar_text" $OUT/summary.txt | head -12 | cut -c1-330
echo "trace + pmc done after $(el)"
# (3) k-mer
KW=$WORK/kmer; mkdir -p $KW
KB="python $REPO/bench_kmer.py --host-queries 0 --repeat 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $KW/trace -o trace -- $KB > $OUT/kmer_trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $KW/pmc_fetch -o pmc -- $KB > $OUT/kmer_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $KW/pmc_write -o pmc -- $KB > $OUT/kmer_pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $KW/pmc_sq -o pmc -- $KB > $OUT/kmer_pmc_sq.log 2>&1
python $REPO/profiles/summarize.py $KW 2>&1 | grep -E "^==|vsx_" > $OUT/kmer_summary.txt
for f in $(find $KW/trace -name "*kernel_stats.csv"); do grep -E "Name|vsx_" $f > $OUT/kmer_kernel_stats.csv; done
python $REPO/profiles/pmc_kmer.py $KW $OUT/pmc_kmer.json > /dev/null 2> $OUT/pmc_kmer.err && cp $OUT/pmc_kmer.json $REPO/profiles/pmc_kmer_current.json
cd $REPO
timeout 300 python bench_kmer.py > $OUT/bench_kmer.json 2> $OUT/bench_kmer.err
echo "kmer rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/bench_kmer.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['kernel_ms'], 'achieved', r['achieved'], 'traffic', r['traffic'], r.get('traffic_GBps'), d['parity_lists_equal_on_sample'])" 2>&1 | cut -c1-300)"
# (4) default line
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$? after $(el)"; python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"), d["end_to_end"].get("ms_calls"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_note"))
s = d.get("search_end_to_end", {})
print({k: s.get(k) for k in ("error", "queries_per_s", "queries_per_s_best", "seconds_later_calls")}, (s.get("reference_cli") or {}).get("same_hits_as_vsx"))
for k, v in d.get("shapes", {}).items(): print(k, v.get("value"), v.get("kernel_split_ms_per_step"), v.get("parity_all_fields_match"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("parity_all_fields_match"))
PY
VSX_DEBUG_TIMING=1 timeout 600 python bench_cluster.py --n 2000000 --parity-prefix 250000 > $OUT/bench_cluster_2M.json 2> $OUT/bench_cluster_2M.err
echo "cluster rc=$? after $(el): $(cut -c1-900 $OUT/bench_cluster_2M.json)"
grep -E "vsx_cluster_fast:" $OUT/bench_cluster_2M.err | tail -1 | cut -c1-400 | tee $OUT/bench_cluster_2M_phases.txt
timeout 900 python bench_allpairs.py --n 50000 --block 1000 --stream 1 --parity-prefix 3000 > $OUT/bench_allpairs_50k.json 2> $OUT/bench_allpairs_50k.err
echo "allpairs 50k rc=$? after $(el): $(python -c "import json; d=json.loads(open('$OUT/bench_allpairs_50k.json').read().strip().splitlines()[-1]); print(d['value'], d['wall_s'], d['align_calls_s'], d['accepted_hits'], d['block_s'][:4], d['parity'])" 2>&1 | cut -c1-600)"
VSX_BENCH_SEARCH_REPS=3 timeout 1500 python bench.py --queries 1250000 --qlen 150 --db 5000000 --dlen 1000 --steps 2 --warmup 1 \
    --no-shapes --ref-search-queries 2048 --e2e-calls 1 > $OUT/config5_share.json 2> $OUT/config5_share.err
echo "config5 rc=$? after $(el)"; python - $OUT/config5_share.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], d["kernel_split_ms_per_step"], "e2e", d.get("value_end_to_end"), d["config"]["workload"][-90:])
    s = d.get("search_end_to_end", {})
    print({k: s.get(k) for k in ("error", "queries_per_s", "seconds", "hits", "vs_reference_cli")}, s.get("reference_cli"))
except Exception as e:
    print("config5 unreadable", e)
PY
PMC=1 bash profiles/run_shapes.sh $TAG/shapes 150x1000x1000000 300x300x400000 400x400x300000 150x300x400000 2>&1 | cut -c1-330 | tee $OUT/shapes.txt
echo "shapes done after $(el)"
bash profiles/run_soaks.sh $TAG/soaks 40 ${SOAK_SEED:-6006} 2>&1 | cut -c1-330 | tee $OUT/soaks.txt
echo "soaks done after $(el)"
bash profiles/shim_cost.sh > $OUT/shim_cost.txt 2>&1; tail -6 $OUT/shim_cost.txt
# (9) RCCL with two ranks on the one GPU of this box
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 180 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend nccl --dry-collectives > $OUT/rccl_two_ranks_one_gpu.txt 2>&1
echo "rccl 2 ranks / 1 GPU rc=$? after $(el): $(grep -E "dry_collectives|Duplicate|rror" $OUT/rccl_two_ranks_one_gpu.txt | head -3 | cut -c1-400)"
echo "all done after $(el)"
