#!/bin/bash
# profiles/pmc_tb.sh [bench args] -- SQ counter passes over one kernels-only bench step; prints the traceback kernels' sums
# (env such as VSX_TB_V1=1 selects the kernel)
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAVES_EQ_64 SQ_IFETCH"; do
  rm -rf /tmp/pq
  rocprofv3 --output-format csv --kernel-trace --pmc $SET -d /tmp/pq -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --kernels-only "$@" > /tmp/pq.log 2>&1
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
dur=collections.defaultdict(float)
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:44]
        if "vsx_t" in k: agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
for k,v in agg.items(): print(k, {c: "%.4g"%x for c,x in v.items()})
if not agg: print(open("/tmp/pq.log").read()[-500:])
PY
done
