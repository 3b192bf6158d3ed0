"""-m gpu: the HIP path (through the C-ABI) against the committed golden vectors (generated from the
reference's own compiled sources) and against the oracle on seeded inputs.  Bit-exact: integer work."""
import os
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


# (r06: (150, 1000) and (300, 300) are the pair shapes of BASELINE configs[4] and [2]: R = 10 against long targets, R = 20 with the
#  half-height traceback tiles -- until now pinned only by bench.py's digests)
@pytest.mark.parametrize("seed,qlen,dlen", [(1, 250, 1000), (2, 150, 300), (3, 400, 400), (4, 64, 64), (5, 300, 80), (6, 150, 1000), (7, 300, 300)])
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


# ---- device-side accept filter (vsx_filter) vs a Python restatement of align_trim + search_acceptable_aligned ----
def _verdict_py(q, t, row, f):
    """core/searchcore.cpp:343-464 (align_trim) + :664-737 (search_acceptable_aligned), evaluated on one result row"""
    import re
    score, al, ma, mi, ga, cigar = row
    runs = [(int(n) if n else 1, op) for n, op in re.findall(r"(\d*)([MID])", cigar)]
    tql = ttl = tqr = ttr = 0
    if runs and runs[0][1] != "M":
        if runs[0][1] == "D": tql = runs[0][0]
        else: ttl = runs[0][0]
    if runs and runs[-1][1] != "M":
        if runs[-1][1] == "D": tqr = runs[-1][0]
        else: ttr = runs[-1][0]
    if tql >= al: tqr = 0
    if ttl >= al: ttr = 0
    indels = al - ma - mi
    ial = al - tql - ttl - tqr - ttr
    iindels = indels - tql - ttl - tqr - ttr
    igaps = ga - (1 if tql + ttl > 0 else 0) - (1 if tqr + ttr > 0 else 0)
    Q, D = len(q), len(t)
    shortest, longest = min(Q, D), max(Q, D)
    iddef = f["iddef"]
    if iddef == 0: idv = 100.0 * ma / shortest if shortest > 0 else 0.0
    elif iddef == 2: idv = 100.0 * ma / ial if ial > 0 else 0.0
    elif iddef == 3: idv = max(0.0, 100.0 * (1.0 - (1.0 * (mi + ga) / longest)))
    else: idv = 100.0 * ma / al if al > 0 else 0.0
    ok = (idv >= 100.0 * f["weak_id"] and mi <= f["maxsubs"] and igaps <= f["maxgaps"] and ial >= f["mincols"]
          and (not f["leftjust"] or tql + ttl == 0) and (not f["rightjust"] or tqr + ttr == 0)
          and ma + mi >= f["query_cov"] * Q and ma + mi >= f["target_cov"] * float(D) and idv <= 100.0 * f["maxid"]
          and (ma + mi > 0 and 100.0 * ma / (ma + mi) >= f["mid"]) and mi + iindels <= f["maxdiffs"])
    if not ok:
        return 3
    return 1 if idv >= 100.0 * f["id"] else 2


@pytest.mark.gpu
@pytest.mark.parametrize("flt", [
    dict(iddef=2, id=0.9, weak_id=0.8),
    dict(iddef=0, id=0.85, weak_id=0.85, maxgaps=2, maxsubs=20),
    dict(iddef=1, id=0.8, weak_id=0.5, mincols=100, maxdiffs=30, query_cov=0.7),
    dict(iddef=3, id=0.7, weak_id=0.6, leftjust=1, target_cov=0.3),
    dict(iddef=4, id=0.95, weak_id=0.9, rightjust=1, maxid=0.99, mid=90.0),
])
def test_device_filter_matches_restatement(gpu_required, flt):
    from vsearch_amd import Aligner
    rng = random.Random(99)
    qs, ts = [], []
    for _ in range(600):
        L = rng.randint(60, 400)
        a = common.rnd_seq(rng, L)
        b = common.mutate(rng, a, rng.choice([0.0, 0.02, 0.05, 0.1, 0.2, 0.35]))
        r = rng.random()
        if r < 0.3:
            b = common.rnd_seq(rng, rng.randint(0, 40)) + b + common.rnd_seq(rng, rng.randint(0, 40))     # terminal gaps
        elif r < 0.5:
            a = common.rnd_seq(rng, rng.randint(0, 30)) + a
        qs.append(a)
        ts.append(b)
    full = dict(iddef=2, id=0.0, weak_id=0.0, maxid=1.0, mid=0.0, query_cov=0.0, target_cov=0.0, maxsubs=2 ** 31 - 1,
                maxgaps=2 ** 31 - 1, mincols=0, maxdiffs=2 ** 31 - 1, leftjust=0, rightjust=0)
    full.update(flt)
    idx = np.arange(len(qs), dtype=np.uint32)
    with Aligner() as al:
        Q, T = al.sequences(qs), al.sequences(ts)
        plain = al.align_pairs(Q, T, idx, idx)
        p = al.plan(Q, T, idx, idx)
        p.set_filter(**full)
        p.run()
        got = p.fetch()
        p.close()
    assert got.verdict is not None and plain.verdict is None
    seen = set()
    for k in range(len(qs)):
        exp = _verdict_py(qs[k], ts[k], plain.row(k), full)
        assert int(got.verdict[k]) == exp, (k, exp, int(got.verdict[k]), plain.row(k))
        seen.add(exp)
        if exp == 3:
            assert got.cigar[k] == "" and got.row(k)[:5] == plain.row(k)[:5]
        else:
            assert got.row(k) == plain.row(k)
    assert len(seen) >= 2, seen


@pytest.mark.gpu
def test_long_targets_leave_column_beyond_int16(gpu_required):
    """targets longer than 32767: the last-row run (right-terminal gap) starts beyond the int16 range of a column index"""
    from oracle import pyoracle
    from vsearch_amd import Aligner
    rng = random.Random(31337)
    orc = pyoracle.Oracle()
    qs, ts = [], []
    for pos in (35000, 60000, 100, 40000):
        q = common.rnd_seq(rng, 90)
        t = common.rnd_seq(rng, pos) + common.mutate(rng, q, 0.03) + common.rnd_seq(rng, rng.randint(0, 4000))
        qs.append(q)
        ts.append(t[:65000])
    idx = np.arange(len(qs), dtype=np.uint32)
    with Aligner() as al:
        res = al.align_pairs(al.sequences(qs), al.sequences(ts), idx, idx)
    for k in range(len(qs)):
        assert res.row(k) == orc.align(qs[k], ts[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("P,nmm", [
    ((4, -7, 7, 7, 22, 10, 18, 4, 5, 2, 4, 6, 5, 5), True),      # ge(query left) 5 > ge(query interior) 4: found by oracle/soak.py
    ((2, -4, 3, 3, 20, 20, 3, 3, 6, 6, 2, 2, 6, 6), False),      # every terminal extension dearer than the interior ones
    ((3, -5, 0, 0, 9, 9, 0, 0, 4, 1, 3, 3, 1, 4), False),
])
def test_terminal_extension_dearer_than_interior(gpu_required, oracle, P, nmm):
    """a scoring set whose left-terminal query-gap extension exceeds the interior one must not take the TOPPAD / tilted
    classes (their dummy rows above the query would offer a cheaper way along the top border): long target overhangs on both
    sides, every field against the oracle (which is the reference's result here, oracle/soak.py)"""
    from vsearch_amd import Aligner
    rng = random.Random(4)
    qs, ts = [], []
    for _ in range(120):
        q = common.rnd_seq(rng, rng.randint(40, 230))
        qs.append(q)
        ts.append(common.rnd_seq(rng, rng.randint(0, 150)) + common.mutate(rng, q, 0.05) + common.rnd_seq(rng, rng.randint(0, 150)))
    qi = np.arange(len(qs), dtype=np.uint32)
    with Aligner(scoring=P, n_mismatch=nmm) as al:
        res = al.align_pairs(al.sequences(qs), al.sequences(ts), qi, qi)
    for k in range(len(qs)):
        assert res.row(k) == tuple(oracle.align(qs[k], ts[k], P, nmm)), (k, qs[k], ts[k])


def test_baseline_shape_vs_the_reference_itself(gpu_required):
    """2 500 queries x 8 family candidates of the BASELINE shape (250 bp vs ~1 kbp): every field incl. the CIGAR against the
    reference's own SSE2 search16 (oracle/_ref/libvsref.so, the reference sources compiled in place)"""
    from oracle import pyoracle
    if not pyoracle.have_ref():
        pytest.fail("oracle/_ref/libvsref.so missing: run `make -C oracle ref` in the build container")
    from vsearch_amd import Aligner
    rng = random.Random(20260924)
    db, fam = common.family_db(rng, 40, 12, 1000, div=0.08)
    qs, src = common.queries_from_db(rng, db, 2500, 250)
    qi, ti = [], []
    for k, s in enumerate(src):
        base = (s // 12) * 12
        for t in rng.sample(range(base, base + 12), 8):
            qi.append(k)
            ti.append(t)
    qi = np.array(qi, np.uint32)
    ti = np.array(ti, np.uint32)
    with Aligner() as al:
        res = al.align_pairs(al.sequences(qs), al.sequences(db), qi, ti)
    ref = pyoracle.Reference()
    try:
        bad = 0
        for k in range(len(qs)):
            rows = ref.search16(qs[k], [db[t] for t in ti[8 * k:8 * k + 8]])
            for x, r in enumerate(rows):
                if tuple(r) != res.row(8 * k + x):
                    bad += 1
                    assert bad < 3, (k, x, tuple(r), res.row(8 * k + x))
        assert bad == 0
    finally:
        ref.close()


@pytest.mark.gpu
def test_chunked_plan_equals_single_chunk(gpu_required, oracle):
    """a checkpoint budget far below the job's needs cuts the plan into many chunks (one buffer, reused chunk after chunk);
    several kernel classes (rows per lane, overflow tracking) are present: results must not depend on the chunking"""
    from vsearch_amd import Aligner
    rng = random.Random(777)
    qs, ts = [], []
    for n, (ql, dl) in enumerate([(60, 200), (250, 900), (400, 400), (700, 1500), (150, 300)] * 30):
        a = common.rnd_seq(rng, ql)
        b = common.rnd_seq(rng, rng.randint(0, dl // 2)) + common.mutate(rng, a, 0.06) + common.rnd_seq(rng, rng.randint(0, dl // 2))
        qs.append(a)
        ts.append(b)
    qs.append(common.rnd_seq(rng, 3000))
    ts.append(common.rnd_seq(rng, 3000) + qs[-1][:500])                 # two strips, overflow-tracked class
    idx = np.arange(len(qs), dtype=np.uint32)
    with Aligner() as al:
        Q, T = al.sequences(qs), al.sequences(ts)
        whole = al.align_pairs(Q, T, idx, idx)
        p = al.plan(Q, T, idx, idx, dir_budget_bytes=1 << 20)
        p.run()
        parts = p.fetch()
        p.run()                                                        # a second run of the same plan reuses the buffer again
        again = p.fetch()
        p.close()
    for k in range(len(qs)):
        assert parts.row(k) == whole.row(k) == again.row(k), k
    for k in range(0, len(qs), 7):
        assert whole.row(k) == oracle.align(qs[k], ts[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"VSX_TRACEBACK": "dirs"}, {"VSX_TB_ARITH": "packed"}, {"VSX_SCORE": "arith"}, {"VSX_NO_SHARE_SUB": "1"}, {"VSX_ROWS": "4"},
    {"VSX_TILT": "0"}, {"VSX_TILT": "0", "VSX_ROWS": "4"}, {"VSX_MAX3": "0"}, {"VSX_MAX3": "0", "VSX_ROWS": "4"},
    {"VSX_SPARSE": "0"}, {"VSX_SPARSE": "0", "VSX_MAX3": "0"},
    # r06: single-strip launches run the ONE variants of the TILT kernels by default; VSX_ONESTRIP=0 keeps the general (multi-strip capable)
    # kernels for every launch -- with whole-wave tasks only, and in the 16-bit TILT class
    {"VSX_ONESTRIP": "0"}, {"VSX_ONESTRIP": "0", "VSX_SPARSE": "0"}, {"VSX_ONESTRIP": "0", "VSX_SPARSE": "0", "VSX_MAX3": "0"},
], ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_alternate_kernel_modes(gpu_required, env):
    """the A/B switches of DESIGN.md section 8 select other kernel variants (stored direction bits, saturating packed traceback,
    table-free scores, unshared subtraction, many strips, plain instead of tilted coordinates, the 16-bit TILT class instead of its
    MAX3 sub-class -- r02's default, which the fixtures otherwise reach only for 1 900 < Q + D < 3 900; r05: every task a whole wave --
    the golden fixtures are one-target tasks, which by default share their waves four at a time): each must reproduce the golden vectors and the torture slice"""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q",
                        "-k", "golden or torture or multi_strip or reference_batch or sparse_task"], env=e, capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


_PIPE_SNIPPET = r"""
import hashlib, random, sys
import numpy as np
sys.path.insert(0, %r)
from tests import common
from vsearch_amd import Aligner
rng = random.Random(21)
fam = [common.rnd_seq(rng, rng.randint(40, 160)) for _ in range(40)]
seqs = [common.mutate(rng, rng.choice(fam), rng.choice([0.03, 0.1, 0.3])) for _ in range(400)] + ["", "A"]
qi = np.array([rng.randrange(len(seqs)) for _ in range(9000)], np.uint32)
qi.sort()
ti = np.array([rng.randrange(len(seqs)) for _ in range(9000)], np.uint32)
h = hashlib.sha256()
with Aligner() as al:
    ss = al.sequences(seqs)
    for flt in (None, dict(id=0.8, weak_id=0.7)):
        res = al.align_pairs_oneshot(ss, ss, qi, ti, filter=flt)
        for arr in (res.score, res.aligned, res.matches, res.mismatches, res.gaps):
            h.update(np.ascontiguousarray(arr).tobytes())
        if flt:
            h.update(np.ascontiguousarray(res.verdict).tobytes())
        h.update("\n".join(res.cigar).encode())
    # the ranked form (filter + per-query order + compaction on the device): slices are cut at query boundaries only
    rk = al.align_pairs_ranked(ss, ss, qi, ti, dict(id=0.8, weak_id=0.7), keep_weak=True)
    for key in ("pair", "score", "aligned", "matches", "mismatches", "gaps", "verdict", "id"):
        h.update(np.ascontiguousarray(rk[key]).tobytes())
    h.update("\n".join(rk["cigar"]).encode())
    h.update(np.sort(rk["undecided"]).tobytes())
    ss.close()
print("HASH", h.hexdigest())
"""


@pytest.mark.gpu
def test_pipelined_slices_equal_one_plan(gpu_required):
    """vsx_align_pairs pipelines large pair lists as several plans (planner thread + GPU + fetch): with the slice size forced
    down to 700 pairs a 9 000-pair call (unfiltered, filtered and ranked, incl. closed-form pairs) must return the same bytes as one
    plan -- with the slices overlapping on the context's two checkpoint blocks and two of them queued behind the fetched one"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({"VSX_PIPELINE": "0"}, {"VSX_PIPELINE_SLICE": "700"}):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", _PIPE_SNIPPET % root], env=e, capture_output=True, text=True, timeout=600, cwd=root)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        outs.append([ln for ln in p.stdout.splitlines() if ln.startswith("HASH")])
    assert outs[0] and outs[0] == outs[1]


def _tilt_reach(P, Q, D):
    """the planner's bound (vsx_host.cpp tilt_possible()): |any value the tilted kernel forms| stays below this"""
    B = max(abs(P[0]), abs(P[1]), *P[8:14])
    G = max(P[2:8])
    return 4 * G + 2 * (Q + ((D + 3) & ~3) + 64) * B


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "nmismatch", "zero_terminal", "uniform10_1"])
def test_tilt_class_boundaries(gpu_required, oracle, name):
    """pairs on both sides of both thresholds of tilt_possible() -- reach < 15 800: MAX3 (values are fp16 bit patterns), < 32 000:
    the 16-bit TILT class, beyond: plain coordinates -- under the four tilt-eligible scoring sets (zero_terminal: no MAX3, its
    right-end gap open is 0); the class each task took is read
    back through vsx_plan_describe, every field of every pair is compared with the oracle.  One query per task (all of a task's
    targets sit on the same side: the planner classifies a task by its longest target)."""
    from vsearch_amd import Aligner
    sc = common.load_golden()["scorings"][name]
    P, nmm = sc["P"], sc["n_mismatch"]
    import zlib
    rng = random.Random(zlib.crc32(name.encode()))
    qs, ts, qi, ti, want = [], [], [], [], []           # want[k] = class of query k's task: 2 MAX3, 1 TILT, 0 plain
    for Q in (250, 400, 90, 520):
        for limit, below, above in ((15800, 2, 1), (32000, 1, 0)):
            # largest padded target length that keeps reach(Q, D) < limit
            D = 8
            while _tilt_reach(P, Q, D + 4) < limit:
                D += 4
            for side, cls in ((0, below), (1, above)):
                if cls == 2 and P[6] <= 0:
                    cls = 1         # MAX3 needs a positive right-end gap open of the query (the kernel's last-row tracking): zero_terminal
                q = common.rnd_seq(rng, Q)
                k = len(qs)
                qs.append(q)
                want.append(cls)
                for x in range(5):
                    d = D - x if side == 0 else D + 1 + x                  # D, D-1, ... stay below; D+1 ... pad to D+4: above
                    assert (_tilt_reach(P, Q, d) < limit) == (side == 0)
                    left = rng.randint(0, d - Q) if d > Q else 0
                    t = (common.rnd_seq(rng, left) + common.mutate(rng, q, 0.08) + common.rnd_seq(rng, d))[:d]
                    qi.append(k)
                    ti.append(len(ts))
                    ts.append(t)
    qi = np.array(qi, np.uint32)
    ti = np.array(ti, np.uint32)
    with Aligner(scoring=P, n_mismatch=nmm) as al:
        Qs, Ts = al.sequences(qs), al.sequences(ts)
        p = al.plan(Qs, Ts, qi, ti)
        info = p.describe()
        p.run()
        res = p.fetch()
        p.close()
    assert info["tasks"] == len(qs)
    assert info["tasks_max3"] == want.count(2), info
    assert info["tasks_tilted"] == want.count(2) + want.count(1), info
    assert info["tasks_tracked"] == 0, info
    for k in range(len(qi)):
        assert res.row(k) == tuple(oracle.align(qs[qi[k]], ts[ti[k]], P, nmm)), (name, k, len(qs[qi[k]]), len(ts[ti[k]]))


@pytest.mark.gpu
def test_sparse_kernel_class_joins_the_dense_one(gpu_required, oracle):
    """r04: a handful of queries a few symbols short of the others would form a kernel class of their own (18 instead of 20 rows per
    lane) and cost a one-wave launch of each kernel; the planner promotes them into the dense class (any row count with 16 R >= Q
    is valid).  One DP launch, and the promoted pairs -- plus a query too short to be promoted -- still equal the oracle."""
    from vsearch_amd import Aligner
    rng = random.Random(404)
    qs = [common.rnd_seq(rng, 300) for _ in range(400)] + [common.rnd_seq(rng, 280) for _ in range(3)] + [common.rnd_seq(rng, 12)]
    ts = [common.mutate(rng, q, 0.06) + common.rnd_seq(rng, rng.randint(0, 30)) for q in qs]
    idx = np.arange(len(qs), dtype=np.uint32)
    with Aligner() as al:
        Q, T = al.sequences(qs), al.sequences(ts)
        p = al.plan(Q, T, idx, idx)
        p.run()
        tm = p.sync()
        res = p.fetch()
        info = p.describe()
        p.close()
    assert info["rows_dominant"] == 20
    assert tm.forward_launches == 2, tm.forward_launches          # the dense class (with the three promoted queries) + the 12-symbol query
    for k in list(range(0, 400, 37)) + [400, 401, 402, 403]:
        assert res.row(k) == tuple(oracle.align(qs[k], ts[k])), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "nmismatch", "zero_terminal", "uniform10_1", "distinct12"])
def test_sparse_task_classes(gpu_required, oracle, name):
    """r05: tasks of <= 2 / <= 4 targets share a wave four / two at a time (vsx_forward_kernel NQ).  Queries of many lengths (several
    row classes, IUPAC symbols, one longer than a strip: never sparse) with 1 .. 7 targets each, the targets of wildly different
    lengths -- so the sub-tasks of a wave have different step ranges, the interior phase ends early for all of them, and the last wave
    of a launch is partly empty -- under the tilt-eligible scoring sets (MAX3 and the 16-bit TILT class) and one that is not (no
    sparse class there).  Every field of every pair against the oracle; the class counts are read back through vsx_plan_describe."""
    from vsearch_amd import Aligner
    doc = common.load_golden()
    sc = doc["scorings"][name]
    P, nmm = sc["P"], sc["n_mismatch"]
    import zlib
    rng = random.Random(zlib.crc32(name.encode()) ^ 0x5A5A)
    qs, ts, qi, ti = [], [], [], []
    ntargets = []
    for rep in range(3):
        for Q in (40, 63, 64, 97, 150, 160, 250, 256, 300, 333, 420, 512, 530):
            q = common.rnd_seq(rng, Q, common.IUPAC + "ACGT" * 6) if rng.random() < 0.25 else common.rnd_seq(rng, Q)
            k = len(qs)
            qs.append(q)
            n = 1 + (k % 7)
            ntargets.append(n)
            for x in range(n):
                shape = rng.random()
                if shape < 0.3:
                    t = common.mutate(rng, q, 0.08)[:max(1, rng.randint(1, Q))]                    # a prefix: shorter than the query
                elif shape < 0.6:
                    t = common.rnd_seq(rng, rng.randint(0, 300)) + common.mutate(rng, q, 0.1) + common.rnd_seq(rng, rng.randint(0, 700))
                elif shape < 0.8:
                    t = common.rnd_seq(rng, rng.randint(1, 24))                                      # a few symbols: the wave's t_switch
                else:
                    t = common.mutate(rng, q, 0.3, "ACGTN")
                qi.append(k)
                ti.append(len(ts))
                ts.append(t)
    qi = np.array(qi, np.uint32)
    ti = np.array(ti, np.uint32)
    with Aligner(scoring=P, n_mismatch=nmm) as al:
        Qs, Ts = al.sequences(qs), al.sequences(ts)
        p = al.plan(Qs, Ts, qi, ti)
        info = p.describe()
        p.run()
        res = p.fetch()
        p.close()
        Qs.close()
        Ts.close()
    if os.environ.get("VSX_SPARSE") == "0" or os.environ.get("VSX_TILT") == "0" or os.environ.get("VSX_TRACEBACK") == "dirs" \
            or os.environ.get("VSX_TB_ARITH") == "packed" or os.environ.get("VSX_SCORE") == "arith" or os.environ.get("VSX_ROWS"):
        pass                                   # (re-run by test_alternate_kernel_modes: other classes, same results)
    elif info["tasks_tilted"] == 0:
        assert info["tasks_sparse"] == 0 and info["waves"] == info["tasks"], info
    else:
        # every query of at most 512 symbols with <= 4 targets whose task took the TILT family is a sparse task
        assert info["tasks_sparse"] > 0 and info["waves"] < info["tasks"], info
        assert info["tasks_sparse"] <= sum(1 for k, n in enumerate(ntargets) if n <= 4 and len(qs[k]) <= 512), info
    for k in range(len(qi)):
        assert res.row(k) == tuple(oracle.align(qs[qi[k]], ts[ti[k]], P, nmm)), (name, k, len(qs[qi[k]]), len(ts[ti[k]]))


_SPARSE_THRESHOLD_SNIPPET = r"""
import random, sys
import numpy as np
sys.path.insert(0, %r)
from tests import common
from vsearch_amd import Aligner
rng = random.Random(8)
seqs = [common.rnd_seq(rng, 150) for _ in range(300)]            # one query length: one row class (the threshold is per kernel class)
out = []
with Aligner() as al:
    S = al.sequences(seqs)
    for n in (600, 9000):
        qi = np.arange(n, dtype=np.uint32) %% 300                      # n one-target tasks (a query may repeat: separate pairs of one query
        qi.sort()                                                       #  group into tasks of <= 8, so use distinct (query, target) draws)
        ti = np.array([rng.randrange(300) for _ in range(n)], np.uint32)
        # one target per task: every pair gets a query of its own
        Qn = al.sequences([seqs[q] for q in qi])
        p = al.plan(Qn, S, np.arange(n, dtype=np.uint32), ti)
        info = p.describe()
        out.append((n, info["tasks"], info["tasks_sparse"], info["waves"]))
        p.close(); Qn.close()
    S.close()
print("INFO", out)
"""


@pytest.mark.gpu
def test_sparse_class_threshold(gpu_required):
    """the production rule (no VSX_SPARSE_MIN in the environment): a sparse-task class below 4 096 tasks stays whole-wave tasks -- one
    launch fewer, nothing to gain on a chip that is not full -- and a larger one shares its waves"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("VSX_SPARSE_MIN", "VSX_SPARSE")}
    p = subprocess.run([sys.executable, "-c", _SPARSE_THRESHOLD_SNIPPET % root], env=e, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("INFO")][-1]
    info = eval(line[5:])
    small, large = info
    assert small[1] == 600 and small[2] == 0 and small[3] == 600, info
    assert large[1] == 9000 and large[2] == 9000 and large[3] == 2250, info


@pytest.mark.gpu
def test_memory_pressure_releases_idle_blocks(gpu_required, oracle):
    """r05: a context keeps its checkpoint blocks, its slab and a pool of idle scratch between plans; whoever runs out of device memory
    calls vsx_internal_memory_pressure(device) and every context of the device gives back what no plan holds (vsx_host.cpp).  Here: two
    contexts with warm blocks, the hook called by hand -- the blocks are gone, bytes were released, and both contexts plan and align again
    (a context with a LIVE plan keeps that plan's block until the plan dies)."""
    import ctypes as C
    from vsearch_amd import Aligner
    lib = gpu_required
    lib.vsx_internal_memory_pressure.argtypes = [C.c_int]
    lib.vsx_internal_memory_pressure.restype = C.c_uint64
    lib.vsx_internal_scratch_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 4)]
    lib.vsx_internal_scratch_sizes.restype = None
    rng = random.Random(31)
    qs = [common.rnd_seq(rng, 200) for _ in range(64)]
    ts = [common.mutate(rng, q, 0.1) + common.rnd_seq(rng, rng.randint(0, 300)) for q in qs]
    idx = np.arange(len(qs), dtype=np.uint32)

    def sizes(al):
        out = (C.c_uint64 * 4)()
        lib.vsx_internal_scratch_sizes(al.h, C.byref(out))
        return list(out)

    with Aligner() as a1, Aligner() as a2:
        Q1, T1 = a1.sequences(qs), a1.sequences(ts)
        Q2, T2 = a2.sequences(qs), a2.sequences(ts)
        first = a1.align_pairs(Q1, T1, idx, idx)
        a2.align_pairs(Q2, T2, idx, idx)
        assert max(sizes(a1)) > 0 and max(sizes(a2)) > 0                  # warm: blocks are kept
        live = a2.plan(Q2, T2, idx, idx)                                  # a2 holds a live plan on one of its blocks
        live.run()
        freed = int(lib.vsx_internal_memory_pressure(0))
        assert freed > 0
        assert sizes(a1) == [0, 0, 0, 0]                                  # a1's blocks went back
        res_live = live.fetch()                                           # the live plan's block survived
        live.close()
        again1 = a1.align_pairs(Q1, T1, idx, idx)
        again2 = a2.align_pairs(Q2, T2, idx, idx)
        for k in range(len(qs)):
            assert first.row(k) == again1.row(k) == again2.row(k) == res_live.row(k), k
        for k in range(0, len(qs), 9):
            assert first.row(k) == tuple(oracle.align(qs[k], ts[k])), k


_PAIR_SNIPPET = r"""
import os, random, sys
import numpy as np
sys.path.insert(0, %r)
from tests import common
from oracle import pyoracle
from vsearch_amd import Aligner
P = %r
nmm = %r
rng = random.Random(97)
orc = pyoracle.Oracle()
qs, ts, qi, ti = [], [], [], []
# queries of several row classes (one multi-strip), plain ACGT except two; 33 .. 75 targets each, some targets with IUPAC symbols
for k, Q in enumerate((120, 150, 250, 256, 300, 333, 400, 640, 200, 260)):
    alpha = "ACGT" if k < 8 else common.IUPAC + "ACGT" * 4
    q = common.rnd_seq(rng, Q, alpha)
    qs.append(q)
    for x in range(33 + 6 * k):
        shape = rng.random()
        if shape < 0.5:
            t = common.rnd_seq(rng, rng.randint(0, 200)) + common.mutate(rng, q, 0.1) + common.rnd_seq(rng, rng.randint(0, 500))
        elif shape < 0.7:
            t = common.mutate(rng, q, 0.25)[:rng.randint(1, Q)]
        elif shape < 0.8:
            t = common.rnd_seq(rng, rng.randint(1, 30))
        else:
            t = common.mutate(rng, q, 0.15, common.IUPAC + "ACGT" * 8)            # may carry N, R, Y ...: such a task stays in the whole-wave class
        qi.append(k); ti.append(len(ts)); ts.append(t)
qi = np.array(qi, np.uint32); ti = np.array(ti, np.uint32)
with Aligner(scoring=P, n_mismatch=nmm) as al:
    Qs, Ts = al.sequences(qs), al.sequences(ts)
    p = al.plan(Qs, Ts, qi, ti, dir_budget_bytes=int(os.environ.get("VSX_TEST_DIR_BUDGET", "0")))
    info = p.describe()
    p.run()
    res = p.fetch()
    p.close()
bad = 0
for k in range(len(qi)):
    if res.row(k) != tuple(orc.align(qs[qi[k]], ts[ti[k]], P, nmm)):
        bad += 1
print("INFO", info["tasks"], info["tasks_pair"], info["tasks_tilted"], bad, info["chunks"])
"""


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "nmismatch", "zero_terminal", "uniform10_1", "distinct12"])
def test_pair_profile_class(gpu_required, name):
    """r05: groups of four whole-wave tasks of one pure-ACGT query with pure-ACGT targets run as a workgroup that shares a pair-indexed
    dword profile (vsx_forward_kernel PAIR; VSX_PAIRPROF=1).  Queries of several row classes with 33 .. 87 targets each, some targets and
    two queries with IUPAC symbols (those tasks stay whole-wave tasks), under the MAX3 class, the 16-bit TILT class and a scoring set that
    cannot be tilted (no pair class there): every field of every pair against the oracle, with the switch on and off."""
    import subprocess
    import sys
    sc = common.load_golden()["scorings"][name]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for mode in ("1", "0"):
        e = dict(os.environ, VSX_PAIRPROF=mode)
        p = subprocess.run([sys.executable, "-c", _PAIR_SNIPPET % (root, tuple(sc["P"]), bool(sc["n_mismatch"]))], env=e, capture_output=True, text=True,
                           timeout=600, cwd=root)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        tasks, pair, tilted, bad, chunks = (int(x) for x in [ln for ln in p.stdout.splitlines() if ln.startswith("INFO")][-1].split()[1:])
        assert bad == 0, (name, mode, bad)
        seen[mode] = (tasks, pair, tilted)
    # r06: the same plan cut into several chunks -- a small checkpoint budget -- so that chunk and launch boundaries fall
    # between PAIR groups of four (never inside one) and the traceback's LDS staging meets first / last tasks of many chunks
    e = dict(os.environ, VSX_PAIRPROF="1", VSX_TEST_DIR_BUDGET=str(24 << 20))
    p = subprocess.run([sys.executable, "-c", _PAIR_SNIPPET % (root, tuple(sc["P"]), bool(sc["n_mismatch"]))], env=e, capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    tasks, pair, tilted, bad, chunks = (int(x) for x in [ln for ln in p.stdout.splitlines() if ln.startswith("INFO")][-1].split()[1:])
    assert bad == 0 and chunks >= 3 and (tasks, pair, tilted) == seen["1"], (name, bad, chunks, tasks, pair, tilted, seen)
    assert seen["0"][1] == 0
    if seen["1"][2] > 0:
        assert seen["1"][1] >= 16 and seen["1"][1] % 4 == 0, seen          # whole groups of four, in several row classes
    else:
        assert seen["1"][1] == 0, seen
