cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk; rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/pk -o t -- python $GRAFT_REPO_ROOT/bench_kmer.py --db 100000 --queries 20000 --host-queries 10 --repeat 1 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:40]
        if "vsx_k" in k: agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
for k,v in agg.items(): print(k, {c: "%.4g"%x for c,x in v.items()})
PY
