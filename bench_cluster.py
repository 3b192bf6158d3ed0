#!/usr/bin/env python3
"""Secondary bench: --cluster_fast (BASELINE config[3] shape, reduced): N x 300 bp amplicons, families of 50 at 2 % divergence,
--id 0.97, sequences sorted by decreasing length, through vsx_cluster_fast.  Prints ONE JSON line.

  python bench_cluster.py [--n 100000 --len 300 --round 4096]
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
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--len", type=int, default=300)
    ap.add_argument("--round", type=int, default=0, help="sequences per GPU round; 0 = the library default (16384)")
    ap.add_argument("--id", type=float, default=0.97)
    ap.add_argument("--rounds", default="", help="A/B: comma-separated round sizes, one vsx_cluster_fast call each on the same searcher "
                                                 "(one short JSON line per call on stderr; the first one is cold, the others find the pools warm)")
    ap.add_argument("--parity-prefix", type=int, default=30000,
                    help="cross-check the S/H records of the first N sequences against the reference CLI run on that prefix (0 = skip)")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_cluster.py needs a GPU (no CPU fallback)")
    from vsearch_amd import Aligner, _lib, workload
    from vsearch_amd._lib import check
    lib = _lib.load()
    dev = torch.device("cuda:0")
    db_ascii, db_off, db_len, fam = workload.make_family_db(a.n, a.len, members=50, div=0.02, seed=31, device=dev)
    host = db_ascii.cpu().numpy()
    del db_ascii
    # --cluster_fast order: length descending (stable), core/db.cpp:433-450; shuffle first so families interleave
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(db_len))
    order = perm[np.argsort(-db_len[perm].astype(np.int64), kind="stable")]
    lens = db_len[order]
    offs = np.zeros(len(lens), np.uint64)
    offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    blob = np.concatenate([host[int(db_off[i]):int(db_off[i]) + int(db_len[i])] for i in order]).tobytes()

    def vp(arr):
        return arr.ctypes.data_as(C.c_void_p)

    with Aligner() as al:
        o = _lib.SearchOpts()
        lib.vsx_search_opts_default(C.byref(o))
        o.id = a.id
        o.maxrejects = 8            # --cluster_fast default (cli.cc:4163-4167); the library default 32 is usearch_global's
        h = C.c_void_p()
        check(lib.vsx_searcher_create(al.h, C.byref(h), C.byref(o), len(lens), C.cast(C.c_char_p(blob), C.c_void_p),
                                      len(blob), vp(offs), vp(lens)), "vsx_searcher_create")
        try:
            for r_ in [int(x) for x in a.rounds.split(",") if x]:
                o2 = _lib.ClusterOut()
                t0 = time.perf_counter()
                check(lib.vsx_cluster_fast(h, r_, C.byref(o2)), "vsx_cluster_fast")
                w2 = time.perf_counter() - t0
                print(json.dumps({"round": r_, "wall_s": round(w2, 3), "clusters": int(o2.n_clusters), "stages": int(o2.hits.stages),
                                  "pairs_aligned": int(o2.hits.pairs_aligned)}), file=sys.stderr, flush=True)
                lib.vsx_cluster_out_free(C.byref(o2))
            out = _lib.ClusterOut()
            t0 = time.perf_counter()
            check(lib.vsx_cluster_fast(h, a.round, C.byref(out)), "vsx_cluster_fast")
            wall = time.perf_counter() - t0
            res = {"metric": "cluster_fast end to end", "value": round(a.n / wall, 1), "unit": "sequences/s", "n_gpus": 1,
                   "higher_is_better": True, "data": "synthetic",
                   "config": {"workload": f"{a.n} x {a.len} bp, families of 50 at 2 % divergence, --id {a.id}, rounds of {a.round or 16384}"},
                   "wall_s": round(wall, 3), "clusters": int(out.n_clusters), "families": int(len(np.unique(fam))),
                   "pairs_aligned": int(out.hits.pairs_aligned), "cells_aligned": int(out.hits.cells_aligned),
                   "seconds_kmer_host": round(out.hits.seconds_kmer, 3), "seconds_align_calls": round(out.hits.seconds_align, 3),
                   "stages": int(out.hits.stages)}
            npfx = min(a.parity_prefix, a.n)
            if npfx > 0:
                # greedy clustering is causal: the records of the first N sequences equal a clustering of that prefix alone
                from oracle import refcli
                if refcli.available():
                    names = [f"s{i:08d}" for i in range(npfx)]          # labels ascending = our order (ties of equal length)
                    seqs = [blob[int(offs[i]):int(offs[i]) + int(lens[i])] for i in range(npfx)]
                    exp, ref_s = refcli.cluster_fast_uc(names, seqs, a.id)
                    exp = [l for l in exp if l[0] in "SH"]
                    cno = np.ctypeslib.as_array(out.clusterno, shape=(a.n,))
                    hfirst = np.ctypeslib.as_array(out.hits.first, shape=(a.n + 1,))
                    cig = C.string_at(out.hits.cigar_blob, int(out.hits.cigar_bytes)) if out.hits.cigar_bytes else b""
                    got = []
                    for s_ in range(npfx):
                        if hfirst[s_] == hfirst[s_ + 1]:
                            got.append(f"S\t{cno[s_]}\t{lens[s_]}\t*\t*\t*\t*\t*\t{names[s_]}\t*")
                        else:
                            hh = out.hits.hit[int(hfirst[s_])]
                            o_ = int(hh.cigar_off)
                            aln = "=" if hh.matches == hh.internal_alignmentlength else cig[o_:cig.index(b"\0", o_)].decode()
                            got.append(f"H\t{cno[s_]}\t{lens[s_]}\t{hh.id:.1f}\t+\t0\t0\t{aln}\t{names[s_]}\t{names[hh.target]}")
                    res["parity"] = {"parity_sample_match": bool(got == exp), "sample": f"S/H records of the first {npfx} sequences vs "
                                     f"vsearch_ref --cluster_fast --uc on that prefix ({sum(1 for l in exp if l[0] == 'H')} H records)",
                                     "reference_s": round(ref_s, 2), "reference_threads": refcli.usable_cpus()}
            lib.vsx_cluster_out_free(C.byref(out))
        finally:
            lib.vsx_searcher_destroy(h)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
