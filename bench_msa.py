#!/usr/bin/env python3
"""Secondary bench (SURVEY.md 8f "next" #3): star MSA + profile + consensus of the clusters of one clustering run
(core/msa.cpp) -- vsx_msa_device_batch (vsx_msa.hip) against the host form vsx_msa on one core (the reference's msa output loop
is single-threaded, cluster.cpp:1449-1530).  Prints ONE JSON line; `bench.py` stays the aligner.

  python bench_msa.py [--clusters 100000 --members 10 --len 300]
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clusters", type=int, default=100_000)
    ap.add_argument("--members", type=int, default=10)
    ap.add_argument("--len", type=int, default=300)
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_msa.py needs a GPU (no CPU fallback)")
    from vsearch_amd import Aligner, _lib
    from vsearch_amd._lib import check
    lib = _lib.load()
    rng = random.Random(3)
    L = a.len
    # a small pool of (member, cigar) variants per centroid pattern keeps the generator cheap; the work is per row anyway
    cen = "".join(rng.choice("ACGT") for _ in range(L))
    variants = []
    for _ in range(64):
        p = rng.randint(10, L - 20)
        kind = rng.random()
        if kind < 0.5:
            variants.append((cen, f"{L}M"))
        elif kind < 0.75:                                      # member lacks 2 centroid positions
            variants.append((cen[:p] + cen[p + 2:], f"{p}M2I{L - p - 2}M"))
        else:                                                  # member has 3 extra symbols
            variants.append((cen[:p] + "TTT" + cen[p:], f"{p}M3D{L - p}M"))
    seqs, cigs, start = [], [], [0]
    for c in range(a.clusters):
        seqs.append(cen.encode()); cigs.append(b"")
        for _ in range(a.members - 1):
            s, g = variants[rng.randrange(64)]
            seqs.append(s.encode()); cigs.append(g.encode())
        start.append(len(seqs))
    n, nc = len(seqs), a.clusters
    sp = (C.c_char_p * n)(*seqs)
    cp = (C.c_char_p * n)(*cigs)
    lens = (C.c_uint32 * n)(*[len(s) for s in seqs])
    st = (C.c_uint64 * (nc + 1))(*start)
    outs = (_lib.MsaOut * nc)()
    with Aligner() as al:
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            check(lib.vsx_msa_device_batch(al.h, nc, st, sp, lens, cp, None, outs), "vsx_msa_device_batch")
            t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
            if rep < 2:
                for k in range(nc):
                    lib.vsx_msa_out_free(C.byref(outs[k]))
        # host, one core, on a sample of clusters; compared with the device's output
        ns = min(nc, 20_000)
        one = _lib.MsaOut()
        same = True
        t0 = time.perf_counter()
        for k in range(ns):
            o = start[k]
            off = lambda arr, ty: C.cast(C.byref(arr, o * C.sizeof(ty)), C.c_void_p)
            check(lib.vsx_msa(start[k + 1] - o, off(sp, C.c_char_p), off(lens, C.c_uint32), off(cp, C.c_char_p), None, C.byref(one)), "vsx_msa")
            d = outs[k]
            nb = int(one.n_rows) * (int(one.alnlen) + 1)
            same = same and one.alnlen == d.alnlen and C.string_at(one.rows, nb) == C.string_at(d.rows, nb) \
                and C.string_at(one.consensus) == C.string_at(d.consensus) \
                and C.string_at(C.cast(one.profile, C.c_void_p), int(one.alnlen) * 48) == C.string_at(C.cast(d.profile, C.c_void_p), int(d.alnlen) * 48)
            lib.vsx_msa_out_free(C.byref(one))
        t_host = time.perf_counter() - t0
        cells = sum((int(outs[k].n_rows) - 1) * int(outs[k].alnlen) for k in range(nc))
        for k in range(nc):
            lib.vsx_msa_out_free(C.byref(outs[k]))
    print(json.dumps({
        "metric": "clusters aligned into star MSA + profile + consensus per second (vsx_msa_device_batch, call time incl. CIGAR parsing, H2D, D2H)",
        "value": round(nc / best, 1), "unit": "clusters/s", "n_gpus": 1, "higher_is_better": True, "dtype": "u8 / u64 counters",
        "data": "synthetic", "config": {"workload": f"{nc} clusters x {a.members} members x {L} bp"},
        "seconds": round(best, 3), "rows_per_s": round(n / best, 1), "row_cells": cells,
        "cpu_baseline": {"value": round(ns / t_host, 1), "unit": "clusters/s", "cores": 1, "kind": "port",
                         "sample": f"first {ns} clusters through vsx_msa (host form of the same algorithm), one core"},
        "parity_equal_on_sample": bool(same),
    }))


if __name__ == "__main__":
    main()
