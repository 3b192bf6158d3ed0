"""profiles/tb_stats.py -- structure of the checkpoint traceback on a bench-shaped workload, from a -DVSX_TB_STATS=1 build
(make -C vsearch_amd/csrc VARIANT=stats EXTRA=-DVSX_TB_STATS=1; VSX_LIBRARY=vsearch_amd/libvsx_stats.so python profiles/tb_stats.py
--qlen 250 --dlen 1000 --db 1000000 --queries 100000): wave iterations, busy (pair, tile) visits, distinct tiles per task and
iteration (= separate sets of checkpoint lines), cells walked."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vsearch_amd import Aligner, SequenceSet, _lib, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=100_000)
ap.add_argument("--qlen", type=int, default=250)
ap.add_argument("--db", type=int, default=1_000_000)
ap.add_argument("--dlen", type=int, default=1000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
db_ascii, db_off, db_len, fam = workload.make_family_db(a.db, a.dlen, seed=17, device=dev)
q_ascii, q_off, q_len, src = workload.make_queries(db_ascii, db_off, db_len, a.queries, a.qlen, seed=11, device=dev)
qidx, tidx = workload.family_candidates(src, fam, per_query=8, seed=5)
torch.cuda.synchronize()
lib = _lib.load()
lib.vsx_internal_tb_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
with Aligner(device=0) as al:
    T = SequenceSet(al, blob=db_ascii.numel(), offsets=db_off, lengths=db_len, device_ptr=db_ascii.data_ptr())
    Q = SequenceSet(al, blob=q_ascii.numel(), offsets=q_off, lengths=q_len, device_ptr=q_ascii.data_ptr())
    plan = al.plan(Q, T, qidx, tidx)
    lib.vsx_internal_tb_stats(out, 1)
    plan.run()
    tm = plan.sync()
    lib.vsx_internal_tb_stats(out, 1)
    info = plan.describe()
    plan.close()
it, busy, distinct, walked, trips, pairs = (int(out[k]) for k in range(6))
print(json.dumps({"shape": [a.qlen, a.dlen], "pairs": pairs, "tasks": info["tasks"], "rows": info["rows_dominant"],
                  "wave_iterations": it, "busy_pair_tiles": busy, "tiles_per_pair": round(busy / max(pairs, 1), 2),
                  "iterations_per_wave": round(it / max(pairs / 64, 1), 2), "lane_utilisation": round(busy / max(it * 64, 1), 3),
                  "distinct_task_tiles": distinct, "pairs_per_distinct_tile": round(busy / max(distinct, 1), 2),
                  "cells_walked": walked, "cells_walked_per_pair_tile": round(walked / max(busy, 1), 2),
                  "traceback_ms_instrumented": round(tm.traceback_ms, 3),
                  "phase_cycles_per_wave_iteration(load1,recompute1,walk1,load2,recompute2,walk2)": [round(int(out[8 + k]) / max(it, 1)) for k in range(6)],
                  "kernel_cycles_per_wave": round(int(out[15]) / max(pairs / 64, 1))}))
