"""Shared helpers for the parity tests: golden fixtures, seeded synthetic data."""
import json
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
IUPAC = "ACGTURYSWKMBDHVN"


def load_golden():
    with open(os.path.join(GOLD, "search16_golden.json")) as f:
        return json.load(f)


def load_api_examples():
    with open(os.path.join(GOLD, "ref_api_examples.json")) as f:
        return json.load(f)


def rnd_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, rate, alphabet="ACGT"):
    """per-base mutation: 80 % substitution, 10 % deletion, 10 % insertion"""
    out = []
    for ch in s:
        r = rng.random()
        if r < rate * 0.8:
            out.append(rng.choice(alphabet))
        elif r < rate * 0.9:
            continue
        elif r < rate:
            out.append(ch)
            out.append(rng.choice(alphabet))
        else:
            out.append(ch)
    return "".join(out)


def family_db(rng, n_families, members, length, div=0.08):
    """family-structured DB: ancestors + members mutated at `div` (a uniform-random DB gives no DP load)"""
    db, fam = [], []
    for f in range(n_families):
        anc = rnd_seq(rng, length)
        for _ in range(members):
            db.append(mutate(rng, anc, div))
            fam.append(f)
    return db, fam


def queries_from_db(rng, db, n, qlen, div=0.03):
    qs, src = [], []
    for _ in range(n):
        k = rng.randrange(len(db))
        m = db[k]
        if len(m) <= qlen:
            w = m
        else:
            o = rng.randrange(len(m) - qlen + 1)
            w = m[o:o + qlen]
        qs.append(mutate(rng, w, div))
        src.append(k)
    return qs, src


def blobify(seqs):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    lens = np.array([len(b) for b in bs], np.uint32)
    off = np.zeros(len(bs), np.uint64)
    if len(bs) > 1:
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    return b"".join(bs), off, lens
