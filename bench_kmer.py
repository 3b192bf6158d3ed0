#!/usr/bin/env python3
"""Secondary bench (SURVEY.md 8f "next" #1): k-mer candidate counting -- the reference's search_topscores
(core/searchcore.cpp:260-340), ~95 % of its wall time at 1 M-sequence databases -- on the device (vsx_kmer.hip) against the
host restatement on the box's CPUs.  `bench.py` (the driver's contract) stays the aligner; this script prints ONE JSON line
of its own:

  python bench_kmer.py [--db 1000000 --dlen 1000 --queries 100000 --qlen 250 --host-queries 1000]

value      = candidate lists per second of the counting kernel (index resident, query words resident)
roofline   = HBM: bytes of postings streamed (the index format decides: 16-byte units of <= 15 packed postings, 2 B per posting in
             the 16-bit format) / kernel time vs 8 TB/s; `lds_atomics` = the second limit of the packed kernel: counter increments
             per clock and CU against the rate a micro-benchmark sustains for random addresses (ubench_lds.hip: 8.1)
cpu_baseline = the host path (vsx_search.cpp candidates_for, all usable cores) on the first --host-queries queries, whose
             lists are also compared with the device's (parity at full database size).
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


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


KMER_SOURCES = ("vsearch_amd/csrc/vsx_kmer.hip", "vsearch_amd/csrc/vsx_kmer.h", "vsearch_amd/csrc/vsx_kmer_pack.h")


def kmer_source_sha():
    import hashlib
    h = hashlib.sha256()
    for f in KMER_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kmer_traffic(a):
    """HBM bytes of the counting kernel per batch from profiles/pmc_kmer_current.json (separate rocprofv3 --pmc passes of THIS command,
    profiles/pmc_kmer.py; refused when the k-mer kernel sources changed since or the workload differs) -> (bytes or None, note)"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_kmer_current.json")))
    except Exception as e:                        # noqa: BLE001
        return None, f"no profiles/pmc_kmer_current.json ({e.__class__.__name__})"
    if d.get("kernel_source_sha") != kmer_source_sha():
        return None, "profiles/pmc_kmer_current.json was measured on other k-mer kernel sources"
    if d.get("workload") != {"queries": a.queries, "qlen": a.qlen, "db": a.db, "dlen": a.dlen}:
        return None, "profiles/pmc_kmer_current.json was measured on another workload"
    return int(d["count_kernel"]["hbm_bytes_per_batch"]), d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--db", type=int, default=1_000_000)
    ap.add_argument("--dlen", type=int, default=1000)
    ap.add_argument("--queries", type=int, default=100_000)
    ap.add_argument("--qlen", type=int, default=250)
    ap.add_argument("--host-queries", type=int, default=1000)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--e2e", type=int, default=0, help="also run vsx_search_batch (--usearch_global --id 0.9) on the first N queries")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_kmer.py needs a GPU (no CPU fallback)")
    from vsearch_amd import Aligner, _lib, workload
    from vsearch_amd._lib import check
    lib = _lib.load()
    dev = torch.device("cuda:0")

    t0 = time.perf_counter()
    db_ascii, db_off, db_len, fam = workload.make_family_db(a.db, a.dlen, seed=17, device=dev)
    q_ascii, q_off, q_len, src = workload.make_queries(db_ascii, db_off, db_len, a.queries, a.qlen, seed=11, device=dev)
    db_blob = db_ascii.cpu().numpy().tobytes()
    q_blob = q_ascii.cpu().numpy().tobytes()
    del db_ascii, q_ascii
    torch.cuda.empty_cache()
    t_gen = time.perf_counter() - t0

    def vp(arr):
        return arr.ctypes.data_as(C.c_void_p)

    out = {}
    with Aligner() as al:
        o = _lib.SearchOpts()
        lib.vsx_search_opts_default(C.byref(o))
        o.id = 0.9
        h = C.c_void_p()
        t0 = time.perf_counter()
        check(lib.vsx_searcher_create(al.h, C.byref(h), C.byref(o), len(db_len), C.cast(C.c_char_p(db_blob), C.c_void_p),
                                      len(db_blob), vp(db_off), vp(db_len)), "vsx_searcher_create")
        t_create = time.perf_counter() - t0
        try:
            def run(device, nq):
                res = _lib.Candidates()
                check(lib.vsx_search_candidates_batch(h, device, nq, C.cast(C.c_char_p(q_blob), C.c_void_p), len(q_blob),
                                                      vp(q_off), vp(q_len), C.byref(res)), "vsx_search_candidates_batch")
                n = int(res.n_queries)
                start = np.ctypeslib.as_array(res.start, shape=(n + 1,)).copy()
                tot = int(start[n])
                tg = np.ctypeslib.as_array(res.target, shape=(max(tot, 1),))[:tot].copy()
                ct = np.ctypeslib.as_array(res.count, shape=(max(tot, 1),))[:tot].copy()
                st = {k: getattr(res, k) for k in ("seconds", "kernel_ms", "index_build_ms", "index_postings", "postings_streamed", "bytes_streamed")}
                lib.vsx_candidates_free(C.byref(res))
                return start, tg, ct, st

            first = None
            best = None
            for r in range(max(1, a.repeat)):
                start, tg, ct, st = run(1, a.queries)
                if first is None:
                    first = st
                if best is None or st["kernel_ms"] < best["kernel_ms"]:
                    best = st
            nh = min(a.host_queries, a.queries)
            if nh > 0:
                run(0, 1)                               # builds the host index (not part of the timed host sample)
                hstart, htg, hct, hst = run(0, nh)
                same = bool(np.array_equal(start[:nh + 1], hstart) and np.array_equal(tg[:hstart[nh]], htg)
                            and np.array_equal(ct[:hstart[nh]], hct))
            else:
                hst, same = {"seconds": float("inf")}, None
            # does the source member lead its query's list? (sanity of the synthetic workload, not a parity statement)
            lead = float(np.mean(tg[start[:-1][start[1:] > start[:-1]]] == src[start[1:] > start[:-1]]))
            e2e = None
            if a.e2e > 0:
                ne = min(a.e2e, a.queries)
                t_first = None
                for rep in range(2):                      # the first call also pays the one-time hipMalloc of the scratch pool
                    hits = _lib.Hits()
                    t0 = time.perf_counter()
                    check(lib.vsx_search_batch(h, ne, C.cast(C.c_char_p(q_blob), C.c_void_p), len(q_blob), vp(q_off), vp(q_len),
                                               C.byref(hits)), "vsx_search_batch")
                    t_e2e = time.perf_counter() - t0
                    if rep == 0:
                        t_first = t_e2e
                        lib.vsx_hits_free(C.byref(hits))
                e2e = {"queries": ne, "seconds_first_call": round(t_first, 3), "seconds": round(t_e2e, 3), "queries_per_s": round(ne / t_e2e, 1),
                       "pairs_aligned": int(hits.pairs_aligned), "cells_aligned": int(hits.cells_aligned), "stages": int(hits.stages),
                       "hits": int(hits.n_hits), "seconds_kmer": round(hits.seconds_kmer, 3), "seconds_align": round(hits.seconds_align, 3)}
                lib.vsx_hits_free(C.byref(hits))
            bytes_streamed = best["bytes_streamed"]                 # what the index format makes the kernel read (packed: 16 B per unit of <= 15 postings)
            gbps = bytes_streamed / (best["kernel_ms"] * 1e-3) / 1e9
            traffic, pmc = kmer_traffic(a)
            out = {
                "metric": "k-mer candidate lists per second (search_topscores: count + threshold, device kernel)",
                "value": round(a.queries / (best["kernel_ms"] * 1e-3), 1), "unit": "queries/s",
                "n_gpus": 1, "higher_is_better": True, "dtype": "u8 counters (u16 for queries with more than 255 words)", "data": "synthetic",
                "config": {"workload": f"{a.queries} x {a.qlen} bp queries vs {a.db} x {a.dlen} bp family DB, wordlength 8, "
                                       "minwordmatches 12 (BASELINE config[1] shape)"},
                "kernel_ms": round(best["kernel_ms"], 3),
                "call_s_incl_host": round(best["seconds"], 3),
                "index": {"build_ms": round(first["index_build_ms"], 1), "postings": int(first["index_postings"]),
                          "bytes": int(first["index_postings"]) * 2},
                "increments_per_s": round(best["postings_streamed"] / (best["kernel_ms"] * 1e-3), 1),
                "bytes_per_posting": round(bytes_streamed / max(1, best["postings_streamed"]), 3),
                "roofline": {"kernel": "vsx_kmer_count_packed_kernel / vsx_kmer_count_kernel (VSX_KMER_PACKED=0)", "bound": "hbm", "achieved": round(gbps, 1), "peak": 8000.0,
                             "unit": "GB/s", "frac": round(gbps / 8000.0, 4), "traffic": traffic,
                             "traffic_note": (pmc if traffic is None else
                                              {"per": "batch of all queries (the count kernel's launches of one vsx_kmer_count_batch call summed)",
                                               "FETCH_SIZE_KiB": pmc["count_kernel"]["fetch_size_kib"], "WRITE_SIZE_KiB": pmc["count_kernel"]["write_size_kib"],
                                               "dispatches": pmc["count_kernel"]["dispatches"], "kernel_source_sha": pmc["kernel_source_sha"],
                                               "SQ": pmc["count_kernel"].get("sq"),
                                               "method": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: KiB units, gfx950 FETCH_SIZE halving)"}),
                             "traffic_GBps": (round(traffic / (best["kernel_ms"] * 1e-3) / 1e9, 1) if traffic else None),
                             "algorithmic_bytes_per_launch": int(bytes_streamed)},
                "lds_atomics": {"per_clock_and_cu": round(best["postings_streamed"] / (best["kernel_ms"] * 1e-3) / (256 * 2.4e9), 2),
                                "measured_random_address_rate": 8.1,
                                "note": "increments only (hops and padding add ~4 %), over the whole timed region incl. the range pre-pass and the selection"},
                "candidates_per_query": round(float(start[-1]) / a.queries, 2),
                "source_member_leads": round(lead, 4),
                "cpu_baseline": {"value": round(nh / hst["seconds"], 1), "unit": "queries/s", "cores": usable_cpus(),
                                 "kind": "port", "sample": f"first {nh} queries, host restatement of search_topscores on all usable cores"},
                "parity_lists_equal_on_sample": same,
                "gen_s": round(t_gen, 1), "create_s": round(t_create, 1),
            }
            if e2e:
                out["usearch_global_end_to_end"] = e2e
        finally:
            lib.vsx_searcher_destroy(h)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
