#!/usr/bin/env python3
"""Secondary bench: --allpairs_global (BASELINE config[4] shape, reduced): N x 400 bp sequences, families of 50 at 10 %
divergence, --id 0.8: every sequence against every later one through vsx_allpairs_block (one GPU plan per block of
queries, device-side accept filter).  Prints ONE JSON line.

  python bench_allpairs.py [--n 5000 --len 400 --block 1000]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--len", type=int, default=400)
    ap.add_argument("--block", type=int, default=1000)
    ap.add_argument("--id", type=float, default=0.8)
    ap.add_argument("--stream", type=int, default=1, help="1 (default): vsx_allpairs_stream -- the blocks' enumeration, alignment and hit completion "
                                                          "overlapped inside the library; 0: one vsx_allpairs_block call per block (r01-r04)")
    ap.add_argument("--max-blocks", type=int, default=0, help="stop after this many blocks (0 = all): timelines of the first, largest blocks")
    ap.add_argument("--parity-prefix", type=int, default=1500,
                    help="cross-check the hits among the first N sequences against the reference CLI (0 = skip)")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_allpairs.py needs a GPU (no CPU fallback)")
    from vsearch_amd import Aligner, _lib, workload
    from vsearch_amd._lib import check
    lib = _lib.load()
    dev = torch.device("cuda:0")
    db_ascii, db_off, db_len, fam = workload.make_family_db(a.n, a.len, members=50, div=0.10, seed=23, device=dev)
    blob = db_ascii.cpu().numpy().tobytes()
    del db_ascii

    def vp(arr):
        return arr.ctypes.data_as(C.c_void_p)

    with Aligner() as al:
        o = _lib.SearchOpts()
        lib.vsx_search_opts_default(C.byref(o))
        o.id = a.id
        h = C.c_void_p()
        check(lib.vsx_searcher_create(al.h, C.byref(h), C.byref(o), len(db_len), C.cast(C.c_char_p(blob), C.c_void_p),
                                      len(blob), vp(db_off), vp(db_len)), "vsx_searcher_create")
        try:
            tot = {"pairs": 0, "cells": 0, "hits": 0, "align_s": 0.0}
            npfx = min(a.parity_prefix, a.n)
            sample = []                                  # userout lines (query, target, id, caln) of the pairs inside the prefix
            per_block = []
            def take(first, hits):
                tot["pairs"] += int(hits.pairs_aligned)
                tot["cells"] += int(hits.cells_aligned)
                tot["hits"] += int(hits.n_hits)
                tot["align_s"] += float(hits.seconds_align)
                if first < npfx and hits.n_hits:
                    harr = np.ctypeslib.as_array(hits.hit, shape=(int(hits.n_hits),))
                    cig = C.string_at(hits.cigar_blob, int(hits.cigar_bytes))
                    for r in harr[(harr["query"] < npfx) & (harr["target"] < npfx)]:
                        o = int(r["cigar_off"])
                        sample.append("t%d\tt%d\t%.1f\t%s" % (r["query"], r["target"], r["id"], cig[o:cig.index(b"\0", o)].decode()))

            t0 = time.perf_counter()
            nrows = a.n if not a.max_blocks else min(a.n, a.max_blocks * a.block)
            if a.stream:
                SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(_lib.Hits))
                last = [time.perf_counter()]

                def sink(_user, first, cnt, hp):
                    try:
                        take(int(first), hp.contents)
                        now = time.perf_counter()
                        per_block.append(round(now - last[0], 3))          # (time between two blocks' hand-overs)
                        last[0] = now
                        return 0
                    except Exception:
                        import traceback
                        traceback.print_exc()
                        return -99
                cb = SINK(sink)
                check(lib.vsx_allpairs_stream(h, 0, 0, nrows, a.block, C.cast(cb, C.c_void_p), None), "vsx_allpairs_stream")
            else:
                for first in range(0, nrows, a.block):
                    cnt = min(a.block, nrows - first)
                    hits = _lib.Hits()
                    tb = time.perf_counter()
                    check(lib.vsx_allpairs_block(h, 0, first, cnt, C.byref(hits)), "vsx_allpairs_block")
                    per_block.append(round(time.perf_counter() - tb, 3))
                    take(first, hits)
                    lib.vsx_hits_free(C.byref(hits))
            wall = time.perf_counter() - t0
        finally:
            lib.vsx_searcher_destroy(h)
    parity = None
    if npfx > 1:
        from oracle import refcli
        if refcli.available():
            seqs = [blob[int(db_off[i]):int(db_off[i]) + int(db_len[i])] for i in range(npfx)]
            exp, ref_s = refcli.allpairs_userout([f"t{i}" for i in range(npfx)], seqs, a.id)
            parity = {"parity_sample_match": bool(sorted(sample) == exp), "sample": f"all {npfx * (npfx - 1) // 2} pairs among the first {npfx} "
                      f"sequences vs vsearch_ref --allpairs_global --userout query+target+id+caln ({len(exp)} accepted pairs)",
                      "reference_s": round(ref_s, 2), "reference_threads": refcli.usable_cpus()}
    # same-family pairs (what --id 0.8 should keep at 10 % divergence from a common ancestor)
    same = int(sum(c * (c - 1) // 2 for c in np.bincount(fam)))
    print(json.dumps({
        "metric": "allpairs_global end to end (pair enumeration + DP + traceback + device filter + accepted hits with CIGAR)",
        "value": round(tot["cells"] / wall / 1e9, 1), "unit": "GCUPS", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
        "config": {"workload": f"{a.n} x {a.len} bp, families of 50 at 10 % divergence, --id {a.id}, blocks of {a.block} queries"},
        "pairs": tot["pairs"], "pairs_per_s": round(tot["pairs"] / wall, 1), "cells": tot["cells"], "wall_s": round(wall, 3),
        "align_calls_s": round(tot["align_s"], 3), "accepted_hits": tot["hits"], "same_family_pairs": same,
        "block_s": per_block, "parity": parity}))


if __name__ == "__main__":
    main()
