"""-m gpu: the HIP path (through the C-ABI) against the committed golden vectors (generated from the
reference's own compiled sources) and against the oracle on seeded inputs.  Bit-exact: integer work."""
import random

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


def _align_cases(aligner_cls, P, nmm, pairs):
    """pairs: list of (q, t) -> list of result rows, through Aligner.align_pairs"""
    with aligner_cls(scoring=P, n_mismatch=nmm) as al:
        qs = al.sequences([q for q, _ in pairs])
        ts = al.sequences([t for _, t in pairs])
        idx = np.arange(len(pairs), dtype=np.uint32)
        res = al.align_pairs(qs, ts, idx, idx)
        rows = [res.row(k) for k in range(len(pairs))]
        qs.close()
        ts.close()
    return rows


def test_golden_vectors(gpu_required):
    """every fixture of tests/golden/search16_golden.json (reference search16 outputs)"""
    from vsearch_amd import Aligner
    doc = common.load_golden()
    by = {}
    for c in doc["cases"]:
        by.setdefault(c["scoring"], []).append(c)
    bad = []
    for name, cases in by.items():
        sc = doc["scorings"][name]
        rows = _align_cases(Aligner, sc["P"], sc["n_mismatch"], [(c["q"], c["t"]) for c in cases])
        for c, r in zip(cases, rows):
            if list(r) != c["exp"]:
                bad.append((name, c["q"][:30], c["t"][:30], c["exp"], r))
    assert not bad, f"{len(bad)} golden mismatches, first: {bad[:3]}"


def test_reference_batch_shape(gpu_required):
    """qprep + search16(seqnos, db): one query against many targets, as align_delayed / allpairs call it"""
    from vsearch_amd import Aligner
    from oracle import pyoracle
    rng = random.Random(5)
    orc = pyoracle.Oracle()
    q = common.rnd_seq(rng, 250)
    db = [common.mutate(rng, q, 0.1) + common.rnd_seq(rng, rng.randint(0, 40)) for _ in range(37)] + ["", "A"]
    with Aligner() as al:
        dbs = al.sequences(db)
        al.qprep(q)
        seqnos = list(range(len(db)))[::-1]
        res = al.search16(seqnos, dbs)
        for k, sn in enumerate(seqnos):
            assert res.row(k) == orc.align(q, db[sn]), (k, sn)
        dbs.close()


@pytest.mark.parametrize("seed,qlen,dlen", [(1, 250, 1000), (2, 150, 300), (3, 400, 400), (4, 64, 64), (5, 300, 80)])
def test_family_pairs_vs_oracle(gpu_required, oracle, seed, qlen, dlen):
    """BASELINE-shaped synthetic pairs (family-structured DB) vs the oracle"""
    from vsearch_amd import Aligner
    rng = random.Random(seed)
    db, fam = common.family_db(rng, 6, 10, dlen)
    qs, src = common.queries_from_db(rng, db, 24, qlen)
    qi, ti = [], []
    for k, s in enumerate(src):
        members = [m for m in range(len(db)) if fam[m] == fam[s]]
        for m in members[:8]:
            qi.append(k)
            ti.append(m)
        qi.append(k)
        ti.append(rng.randrange(len(db)))       # plus an unrelated target
    with Aligner() as al:
        Q, T = al.sequences(qs), al.sequences(db)
        res = al.align_pairs(Q, T, qi, ti)
    qb, qo, ql = common.blobify(qs)
    tb, to, tl = common.blobify(db)
    sc, a, m, mm, g, cig = oracle.align_batch(qb, qo, ql, tb, to, tl, qi, ti)
    assert np.array_equal(res.score, sc)
    assert np.array_equal(res.aligned, a)
    assert np.array_equal(res.matches, m)
    assert np.array_equal(res.mismatches, mm)
    assert np.array_equal(res.gaps, g)
    assert res.cigar == cig


def test_torture_slice(gpu_required, oracle):
    """IUPAC codes, lower case, unknown symbols, length-1 and ragged lengths, all scoring sets"""
    from vsearch_amd import Aligner
    doc = common.load_golden()
    rng = random.Random(77)
    pairs = []
    for _ in range(300):
        kind = rng.random()
        if kind < 0.3:
            a = common.rnd_seq(rng, rng.randint(1, 90), common.IUPAC + "acgtn-X")
            b = common.mutate(rng, a, 0.15, common.IUPAC + "acgtn")
        elif kind < 0.5:
            a, b = common.rnd_seq(rng, rng.randint(1, 3)), common.rnd_seq(rng, rng.randint(1, 40))
        elif kind < 0.7:
            a = common.rnd_seq(rng, rng.randint(1, 200))
            b = common.mutate(rng, a, 0.3, "ACGTN")
        else:
            a, b = common.rnd_seq(rng, rng.randint(0, 70)), common.rnd_seq(rng, rng.randint(0, 130))
        pairs.append((a, b))
    for name, sc in doc["scorings"].items():
        rows = _align_cases(Aligner, sc["P"], sc["n_mismatch"], pairs)
        for (a, b), r in zip(pairs, rows):
            assert r == oracle.align(a, b, sc["P"], sc["n_mismatch"]), (name, a, b)


def test_long_queries_multi_strip(gpu_required, oracle):
    """queries longer than one 16-lane strip (Q > 512) and long targets"""
    from vsearch_amd import Aligner
    rng = random.Random(9)
    pairs = []
    for qlen, dlen in [(513, 300), (700, 700), (1500, 200), (2000, 1800), (33, 3000), (4000, 900)]:
        a = common.rnd_seq(rng, qlen)
        b = common.mutate(rng, a, 0.1)[:dlen] if dlen <= qlen else common.mutate(rng, a, 0.1) + common.rnd_seq(rng, dlen - qlen)
        pairs.append((a, b))
    rows = _align_cases(Aligner, (2, -4, 1, 1, 18, 18, 1, 1, 1, 1, 2, 2, 1, 1), False, pairs)
    for (a, b), r in zip(pairs, rows):
        assert r == oracle.align(a, b), (len(a), len(b), r[:5])


def test_size_guard_and_sentinels(gpu_required):
    """Q+D > 65535 or Q*D > 25e6 -> sentinel without DP (align_simd.cpp:130-134)"""
    from vsearch_amd import Aligner
    with Aligner() as al:
        assert al.align("A" * 5001, "C" * 5000) == (32767, 0, 0, 0, 0, "")
        assert al.align("A" * 10, "C" * 65530) == (32767, 0, 0, 0, 0, "")
        assert al.align("ACGT", "") == (32767, 0, 0, 0, 0, "")
        assert al.align("", "") == (0, 0, 0, 0, 0, "")
        assert al.align("", "ACGT") == (-5, 4, 0, 0, 4, "4I")
        assert al.align("", "A") == (-2, 1, 0, 0, 1, "1I")
