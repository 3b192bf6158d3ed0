#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: randomized soak of the HIP aligner against the REFERENCE's own search16 (oracle/_ref/libvsref.so).

Every round draws a scoring set (match / mismatch / the twelve gap penalties all independent, n_mismatch at random; now and then
values large enough to reach the 16-bit limits, the forced-fallback threshold and zero penalties), a pair population of one
shape class (related / unrelated / IUPAC + lower case / tiny / gappy / long-target / square) and compares every field of every
pair -- score (incl. the SHRT_MAX sentinel), aligned, matches, mismatches, gaps, CIGAR.  Runs on the GPU box:

    python oracle/soak.py --seconds 120 --seed 1 --out gpurun_out/soak.json
"""
import argparse
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402
from oracle.gen_golden import IUPAC, make_pair, mutate, rnd_seq  # noqa: E402


def draw_scoring(rng):
    kind = rng.random()
    if kind < 0.15:
        return pyoracle.DEFAULT_P, rng.random() < 0.3, "default"
    if kind < 0.25:     # values that push H towards the 16-bit limits / the sentinel
        m = rng.choice([40, 120, 300])
        return (m, -rng.choice([60, 400, 3000])) + tuple(rng.choice([1, 30, 200, 2000]) for _ in range(6)) + tuple(rng.choice([1, 20, 300]) for _ in range(6)), False, "large"
    if kind < 0.32:     # at / beyond the clamps (forced fallback: every pair answers the sentinel)
        P = [2, -4] + [rng.choice([6553, 6554, 1, 18]) for _ in range(6)] + [rng.choice([1, 2, 6553]) for _ in range(6)]
        return tuple(P), False, "clamp"
    match = rng.randint(1, 6)
    mism = -rng.randint(1, 9)
    opens = [rng.randint(0, 30) for _ in range(6)]
    exts = [rng.randint(0, 6) for _ in range(6)]
    if rng.random() < 0.3:          # the common symmetric case: query and target penalties equal
        opens = [opens[0], opens[0], opens[2], opens[2], opens[4], opens[4]]
        exts = [exts[0], exts[0], exts[2], exts[2], exts[4], exts[4]]
    return (match, mism) + tuple(opens) + tuple(exts), rng.random() < 0.3, "random"


def draw_population(rng, nq, nt):
    """nq queries x nt targets each (one reference search16 call per query), one shape class per round"""
    shape = rng.choice(["related", "unrelated", "iupac", "tiny", "gappy", "long_target", "square", "mixed", "multi_strip"])
    if shape == "multi_strip":      # queries beyond 16 lanes x 32 rows: several strips, handed over through HBM; few, they are big
        nq = max(2, nq // 16)
    qs, ts = [], []
    for _ in range(nq):
        if shape in ("related", "unrelated", "iupac", "tiny", "gappy"):
            q, t0 = make_pair(rng, shape)
            tt = [t0] + [mutate(rng, t0, rng.choice([0.02, 0.1])) if t0 else "" for _ in range(nt - 1)]
        elif shape == "long_target":
            q = rnd_seq(rng, rng.randint(30, 260))
            tt = []
            for _ in range(nt):
                core = mutate(rng, q, rng.choice([0.03, 0.1]))
                tt.append(rnd_seq(rng, rng.randint(0, 700)) + core + rnd_seq(rng, rng.randint(0, 700)))
        elif shape == "multi_strip":
            q = rnd_seq(rng, rng.randint(513, 2600))
            tt = []
            for _ in range(nt):
                core = mutate(rng, q, rng.choice([0.02, 0.08]))
                if rng.random() < 0.5:
                    a = rng.randint(0, len(core) // 2)
                    core = core[a:a + rng.randint(200, len(core))]
                tt.append(rnd_seq(rng, rng.randint(0, 300)) + core + rnd_seq(rng, rng.randint(0, 300)))
        elif shape == "square":
            L = rng.randint(200, 520)
            q = rnd_seq(rng, L)
            tt = [mutate(rng, q, rng.choice([0.02, 0.05, 0.15])) for _ in range(nt)]
        else:
            q, _ = make_pair(rng, rng.choice(["related", "iupac", "gappy"]))
            tt = [make_pair(rng, rng.choice(["related", "unrelated", "tiny", "iupac"]))[1] for _ in range(nt)]
        if rng.random() < 0.1:
            q = "".join(c.lower() if rng.random() < 0.3 else c for c in q)
        qs.append(q)
        ts.append(tt)
    return shape, qs, ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--queries", type=int, default=160)
    ap.add_argument("--targets", type=int, default=8)
    ap.add_argument("--out", default="")
    ap.add_argument("--max-rounds", type=int, default=0, help="stop after this many rounds (0: run for --seconds): a deterministic set of rounds for a given seed")
    a = ap.parse_args()
    if not pyoracle.have_ref():
        raise SystemExit("oracle/_ref/libvsref.so missing: make -C oracle ref")
    from vsearch_amd import Aligner
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    rounds = pairs = bad = sentinels = cells = 0
    by_kind, by_shape, examples, failing = {}, {}, [], []
    while time.time() < t_end and (a.max_rounds <= 0 or rounds < a.max_rounds):
        P, nmm, kind = draw_scoring(rng)
        shape, qs, ts = draw_population(rng, a.queries, a.targets)
        flat = [t for tt in ts for t in tt]
        qi = np.repeat(np.arange(len(qs), dtype=np.uint32), a.targets)
        ti = np.arange(len(flat), dtype=np.uint32)
        with Aligner(scoring=P, n_mismatch=nmm) as al:
            res = al.align_pairs(al.sequences(qs), al.sequences(flat), qi, ti)
        ref = pyoracle.Reference(P, nmm)
        bad_before = bad
        try:
            for k, q in enumerate(qs):
                rows = ref.search16(q, ts[k])
                for x, r in enumerate(rows):
                    got = res.row(k * a.targets + x)
                    pairs += 1
                    cells += len(q) * len(ts[k][x])
                    sentinels += r[0] == 32767
                    if tuple(r) != got:
                        bad += 1
                        if len(examples) < 12 and (bad - bad_before) <= 2:
                            examples.append({"P": list(P), "n_mismatch": nmm, "q": q, "t": ts[k][x], "ref": list(r), "hip": list(got)})
        finally:
            ref.close()
        if bad > bad_before and len(failing) < 40:
            failing.append({"P": list(P), "n_mismatch": nmm, "shape": shape, "kind": kind, "mismatches": bad - bad_before})
        rounds += 1
        by_kind[kind] = by_kind.get(kind, 0) + 1
        by_shape[shape] = by_shape.get(shape, 0) + 1
    out = {"rounds": rounds, "pairs": pairs, "cells": cells, "mismatches": bad, "sentinel_pairs": sentinels, "scoring_kinds": by_kind,
           "shapes": by_shape, "seed": a.seed, "seconds": a.seconds, "failing_rounds": failing, "examples": examples,
           "what": "HIP aligner (vsx_align_pairs) vs the reference's search16 (oracle/_ref/libvsref.so): all six output fields of every pair"}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
