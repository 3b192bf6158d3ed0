"""-m gpu: the filters and orderings of the dispatch layer that depend on abundances / labels / options beyond --id,
each against the reference CLI itself (oracle/_ref/vsearch_ref):

  search_acceptable_unaligned  maxqsize, mintsize, minsizeratio, maxsizeratio, --self   core/searchcore.cpp:541-609
  search_acceptable_aligned    UNOISE skew rule (--cluster_unoise)                      core/searchcore.cpp:701-718
  hit_compare_bysize / search_findbest2_bysize (--cluster_size --sizeorder)             core/searchcore.cpp:182-243, :994-1025
  accept_verdict() on the device (vsx_filter) == align_trim + search_acceptable_aligned of the reference, pair by pair
"""
import os
import random
import subprocess

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vsearch_ref")
FIELDS = ["query", "target", "id", "alnlen", "mism", "opens", "exts", "raw", "caln", "id0", "id1", "id2", "id3", "id4"]


def _need_ref():
    if not os.path.exists(REF_BIN):
        pytest.fail("oracle/_ref/vsearch_ref missing: run `make -C oracle ref_full` in the build container")


def _run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return p


def _first_diff(got, exp):
    for i, (a, b) in enumerate(zip(got, exp)):
        if a != b:
            return f"line {i}:\n got {a}\n exp {b}"
    return f"length {len(got)} vs {len(exp)}"


def _write(path, names, seqs):
    with open(path, "w") as f:
        f.write("".join(f">{n}\n{s}\n" for n, s in zip(names, seqs)))


@pytest.mark.parametrize("name,opts,extra", [
    ("ratios", dict(id=0.9, maxaccepts=3, maxsizeratio=2.0, minsizeratio=0.25, mintsize=2, maxqsize=50),
     ["--id", "0.9", "--maxaccepts", "3", "--maxsizeratio", "2.0", "--minsizeratio", "0.25", "--mintsize", "2", "--maxqsize", "50"]),
    ("maxsizeratio_only", dict(id=0.85, maxaccepts=0, maxrejects=0, maxsizeratio=0.5),
     ["--id", "0.85", "--maxaccepts", "0", "--maxrejects", "0", "--maxsizeratio", "0.5"]),
])
def test_abundance_filters_match_reference_cli(gpu_required, tmp_path, name, opts, extra):
    """--usearch_global --sizein with the abundance filters of search_acceptable_unaligned"""
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(3)
    db, fam = common.family_db(rng, 20, 10, 400, div=0.05)
    tsize = [rng.choice([1, 1, 2, 3, 5, 8, 20, 100]) for _ in db]
    qs, _ = common.queries_from_db(rng, db, 80, 200)
    qsize = [rng.choice([1, 2, 4, 10, 60]) for _ in qs]
    tn = [f"t{i};size={tsize[i]}" for i in range(len(db))]
    qn = [f"q{i};size={qsize[i]}" for i in range(len(qs))]
    tmp = str(tmp_path)
    _write(tmp + "/db.fa", tn, db)
    _write(tmp + "/q.fa", qn, qs)
    _run([REF_BIN, "--usearch_global", tmp + "/q.fa", "--db", tmp + "/db.fa", "--qmask", "none", "--dbmask", "none", "--threads", "1",
          "--userout", tmp + "/u.tsv", "--userfields", "+".join(FIELDS), "--quiet", "--sizein"] + extra)
    exp = open(tmp + "/u.tsv").read().splitlines()
    with Aligner() as al:
        ss = SearchSession(al, db, sizes=tsize, labels=tn, **opts)
        hits = ss.search_batch(qs, sizes=qsize, labels=qn)
        got = ss.userout(qs, qnames=qn, tnames=tn, fields=FIELDS, hits=hits)
        # the same searcher without abundances must report MORE (the filters really fired)
        ss2 = SearchSession(al, db, **{k: v for k, v in opts.items() if k in ("id", "maxaccepts", "maxrejects")})
        plain = ss2.userout(qs, qnames=qn, tnames=tn, fields=FIELDS)
    assert len(exp) > 20
    assert got == exp, _first_diff(got, exp)
    assert len(plain) > len(got)


def test_self_filter_matches_reference_cli(gpu_required, tmp_path):
    """--self: a target carrying the query's own label is rejected before alignment (searchcore.cpp:600-602)"""
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(8)
    db, fam = common.family_db(rng, 12, 8, 300, div=0.04)
    names = [f"seq{i}" for i in range(len(db))]
    pick = list(range(0, len(db), 3))
    qs, qn = [db[i] for i in pick], [names[i] for i in pick]
    tmp = str(tmp_path)
    _write(tmp + "/db.fa", names, db)
    _write(tmp + "/q.fa", qn, qs)
    extra = ["--id", "0.9", "--maxaccepts", "2", "--self"]
    _run([REF_BIN, "--usearch_global", tmp + "/q.fa", "--db", tmp + "/db.fa", "--qmask", "none", "--dbmask", "none", "--threads", "1",
          "--userout", tmp + "/u.tsv", "--userfields", "+".join(FIELDS), "--quiet"] + extra)
    exp = open(tmp + "/u.tsv").read().splitlines()
    with Aligner() as al:
        ss = SearchSession(al, db, labels=names, id=0.9, maxaccepts=2, self_=1)
        hits = ss.search_batch(qs, labels=qn)
        got = ss.userout(qs, qnames=qn, tnames=names, fields=FIELDS, hits=hits)
    assert len(exp) > 10 and all(l.split("\t")[0] != l.split("\t")[1] for l in exp)
    assert got == exp, _first_diff(got, exp)


def _amplicons(rng):
    seqs = []
    for f in range(20):
        anc = common.rnd_seq(rng, 300)
        for _ in range(rng.randint(3, 12)):
            seqs.append(common.mutate(rng, anc, rng.choice([0.005, 0.01, 0.02])))
    sz = [rng.choice([1, 1, 1, 2, 3, 5, 9, 30, 200]) for _ in seqs]
    names = [f"s{i:04d};size={sz[i]}" for i in range(len(seqs))]
    # Database::sortbyabundance (core/db.cpp:471-486): abundance descending, then label, then input order
    order = sorted(range(len(seqs)), key=lambda i: (-sz[i], names[i], i))
    return seqs, sz, names, order


def test_cluster_size_sizeorder_matches_reference_cli(gpu_required, tmp_path):
    """--cluster_size --sizein --sizeorder --maxaccepts 4: a member joins the most abundant accepted centroid
    (search_findbest2_bysize), not the most similar one"""
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(3)
    seqs, sz, names, order = _amplicons(rng)
    tmp = str(tmp_path)
    _write(tmp + "/c.fa", names, seqs)
    for so, flag in ((1, ["--sizeorder"]), (0, [])):
        _run([REF_BIN, "--cluster_size", tmp + "/c.fa", "--qmask", "none", "--threads", "1", "--uc", tmp + "/c.uc", "--quiet",
              "--id", "0.97", "--sizein", "--maxaccepts", "4"] + flag)
        exp = open(tmp + "/c.uc").read().splitlines()
        sseqs, snames, ssz = [seqs[i] for i in order], [names[i] for i in order], [sz[i] for i in order]
        with Aligner() as al:
            ss = SearchSession(al, sseqs, sizes=ssz, labels=snames, id=0.97, maxaccepts=4, maxrejects=32, sizeorder=so)
            got = ss.uc_lines(snames, round=37, sizes=ssz, command="cluster_size")
        assert sum(1 for l in exp if l[0] == "H") > 40
        assert got == exp, (so, _first_diff(got, exp))
        if so:
            by_size = exp
    # the option must matter on this input, otherwise the test pins nothing
    assert by_size != exp


def test_cluster_unoise_matches_reference_cli(gpu_required, tmp_path):
    """--cluster_unoise: accept on the skew rule beta(d) = 1 / 2^(alpha d + 1) instead of --id (searchcore.cpp:701-718),
    weak_id forced to 0.90 (cli.cc:4153)"""
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(4)
    seqs, sz, names, order = _amplicons(rng)
    tmp = str(tmp_path)
    _write(tmp + "/c.fa", names, seqs)
    _run([REF_BIN, "--cluster_unoise", tmp + "/c.fa", "--qmask", "none", "--threads", "1", "--uc", tmp + "/n.uc", "--quiet",
          "--minsize", "1", "--sizein", "--unoise_alpha", "2.0"])
    exp = open(tmp + "/n.uc").read().splitlines()
    sseqs, snames, ssz = [seqs[i] for i in order], [names[i] for i in order], [sz[i] for i in order]
    with Aligner() as al:
        # cluster_unoise runs with the search defaults: --id is not given (the rule replaces it), maxrejects 32
        ss = SearchSession(al, sseqs, sizes=ssz, labels=snames, id=0.0, maxaccepts=1, maxrejects=32, cluster_unoise=1, unoise_alpha=2.0)
        got = ss.uc_lines(snames, round=29, sizes=ssz, command="cluster_unoise")
    assert sum(1 for l in exp if l[0] == "H") > 20 and sum(1 for l in exp if l[0] == "S") > 30
    assert got == exp, _first_diff(got, exp)


FILTERS = [
    (dict(iddef=2, id=0.9, weak_id=0.8), ["--iddef", "2"]),
    (dict(iddef=0, id=0.85, weak_id=0.85, maxgaps=2, maxsubs=20), ["--iddef", "0", "--maxgaps", "2", "--maxsubs", "20"]),
    (dict(iddef=1, id=0.8, weak_id=0.5, mincols=100, maxdiffs=30, query_cov=0.7),
     ["--iddef", "1", "--mincols", "100", "--maxdiffs", "30", "--query_cov", "0.7"]),
    (dict(iddef=3, id=0.7, weak_id=0.6, leftjust=1, target_cov=0.3), ["--iddef", "3", "--leftjust", "--target_cov", "0.3"]),
    (dict(iddef=4, id=0.95, weak_id=0.9, rightjust=1, maxid=0.99, mid=90.0), ["--iddef", "4", "--rightjust", "--maxid", "0.99", "--mid", "90.0"]),
]


@pytest.mark.parametrize("flt,extra", FILTERS, ids=[f"iddef{f['iddef']}" for f, _ in FILTERS])
def test_device_filter_matches_reference_cli(gpu_required, tmp_path, flt, extra):
    """accept_verdict() in the traceback kernel against the reference's own align_trim + search_acceptable_aligned: every
    pair (i < j) of a sequence set goes through --allpairs_global of the reference CLI twice -- with --id = id (its output
    = the ACCEPTED pairs) and with --id = weak_id (= ACCEPTED + WEAK) -- and through vsx_align_pairs_filtered once."""
    _need_ref()
    from vsearch_amd import Aligner
    rng = random.Random(99)
    seqs = []
    for f in range(6):
        anc = common.rnd_seq(rng, rng.randint(120, 380))
        for _ in range(7):
            b = common.mutate(rng, anc, rng.choice([0.0, 0.02, 0.05, 0.1, 0.2]))
            r = rng.random()
            if r < 0.3:
                b = common.rnd_seq(rng, rng.randint(0, 40)) + b + common.rnd_seq(rng, rng.randint(0, 40))     # terminal gaps
            elif r < 0.5:
                b = b[rng.randint(0, 30):]
            seqs.append(b)
    n = len(seqs)
    names = [f"t{i}" for i in range(n)]
    tmp = str(tmp_path)
    _write(tmp + "/a.fa", names, seqs)

    def ref_pairs(idv):
        _run([REF_BIN, "--allpairs_global", tmp + "/a.fa", "--qmask", "none", "--threads", "1", "--userout", tmp + "/ua.tsv",
              "--userfields", "query+target+caln", "--quiet", "--id", repr(idv)] + extra)
        return {(int(l.split("\t")[0][1:]), int(l.split("\t")[1][1:])): l.split("\t")[2] for l in open(tmp + "/ua.tsv").read().splitlines()}
    accepted = ref_pairs(flt["id"])
    passing = ref_pairs(flt["weak_id"])
    assert set(accepted) <= set(passing)
    qi, ti = np.triu_indices(n, 1)
    with Aligner() as al:
        S = al.sequences(seqs)
        got = al.align_pairs_oneshot(S, S, qi.astype(np.uint32), ti.astype(np.uint32), filter=flt)
    assert got.verdict is not None
    seen = set()
    for k in range(len(qi)):
        key = (int(qi[k]), int(ti[k]))
        exp = 1 if key in accepted else (2 if key in passing else 3)
        assert int(got.verdict[k]) == exp, (key, exp, int(got.verdict[k]), got.row(k))
        seen.add(exp)
        if exp == 3:
            assert got.cigar[k] == ""
        else:
            assert got.cigar[k] == passing[key], key
    assert len(seen) >= 2, seen


def test_device_reverse_complement_matches_reference_cli(gpu_required, tmp_path):
    """vsx_seqset_create_both_strands (reverse complement in the 4-bit code domain on the device) against the reference's own
    --fastx_revcomp (utils/reverse_complement.cpp + chrmap_complement): aligning the device-made minus strand of q with t must
    equal aligning the reference's reverse-complemented text of q with t -- IUPAC, lower case and U included."""
    _need_ref()
    from vsearch_amd import Aligner
    rng = random.Random(12)
    qs = [common.rnd_seq(rng, rng.randint(1, 300), "ACGT") for _ in range(20)]
    qs += [common.rnd_seq(rng, rng.randint(20, 200), "ACGTURYSWKMBDHVNacgtunryk") for _ in range(30)]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    ts = []
    for q in qs:                                     # targets related to the MINUS strand, so that the alignments are not trivial
        core = "".join(comp.get(c.upper(), "N") for c in reversed(q))
        ts.append(common.mutate(rng, core, 0.05, "ACGTN") + common.rnd_seq(rng, rng.randint(0, 30)))
    tmp = str(tmp_path)
    _write(tmp + "/q.fa", [f"q{i}" for i in range(len(qs))], qs)
    _run([REF_BIN, "--fastx_revcomp", tmp + "/q.fa", "--fastaout", tmp + "/rc.fa", "--fasta_width", "0", "--quiet"])
    rc = [l for l in open(tmp + "/rc.fa").read().splitlines() if not l.startswith(">")]
    assert len(rc) == len(qs) and any(set(r) - set("ACGTN") for r in rc)
    n = len(qs)
    idx = np.arange(n, dtype=np.uint32)
    with Aligner() as al:
        both = al.sequences(qs, both_strands=True)
        assert len(both) == 2 * n
        T = al.sequences(ts)
        dev = al.align_pairs(both, T, np.concatenate([idx, idx + n]), np.concatenate([idx, idx]))
        ref_plus = al.align_pairs(al.sequences(qs), T, idx, idx)
        ref_minus = al.align_pairs(al.sequences(rc), T, idx, idx)
    for k in range(n):
        assert dev.row(k) == ref_plus.row(k), k
        assert dev.row(n + k) == ref_minus.row(k), (k, qs[k], rc[k])
    assert sum(1 for k in range(n) if ref_minus.row(k)[2] > 10) > 20


@pytest.mark.parametrize("keep_weak", [False, True])
def test_device_ranking_equals_host_sort(gpu_required, keep_weak):
    """vsx_align_pairs_ranked (flag + scan + per-query stable sort by identity + gather on the device, vsx_rank.hip) against a
    host sort of the full vsx_align_pairs_filtered result: same kept set, same order (query, id descending, pair order), same
    fields (the pipelined form of the same path runs in tests/test_gpu_scale.py: 2.0 M pairs)."""
    from vsearch_amd import Aligner
    rng = random.Random(7)
    db, fam = common.family_db(rng, 12, 10, 260, div=0.06)
    db += [common.rnd_seq(rng, 250) for _ in range(20)]
    db.append(common.rnd_seq(rng, 40000))                           # 250 nt against 40 kb: the 16-bit DP overflows AT RUN TIME (:1774-1786)
    db.append("")                                                   # an empty target: sentinel pairs answered before the launch
    n = len(db)
    flt = dict(iddef=2, id=0.85, weak_id=0.7)
    qi, ti = np.triu_indices(n, 1)
    qi, ti = qi.astype(np.uint32), ti.astype(np.uint32)
    with Aligner() as al:
        S = al.sequences(db)
        full = al.align_pairs_oneshot(S, S, qi, ti, filter=flt)
        rk = al.align_pairs_ranked(S, S, qi, ti, flt, keep_weak=keep_weak)
    keep = [k for k in range(len(qi)) if full.verdict[k] == 1 or (keep_weak and full.verdict[k] == 2)]
    # identity of the filter (iddef 2) recomputed from the row is not needed: the order key is (query, -id, pair) and the id
    # comes back from the device; check it is consistent with matches / internal length and monotone inside every query
    assert sorted(rk["pair"].tolist()) == keep
    assert len(keep) > 300
    pos = {int(p): j for j, p in enumerate(rk["pair"])}
    for k in keep:
        j = pos[k]
        assert (int(rk["score"][j]), int(rk["aligned"][j]), int(rk["matches"][j]), int(rk["mismatches"][j]), int(rk["gaps"][j]),
                rk["cigar"][j]) == full.row(k), k
        assert int(rk["verdict"][j]) == int(full.verdict[k])
    order = sorted(keep, key=lambda k: (int(qi[k]), -float(rk["id"][pos[k]]), k))
    assert rk["pair"].tolist() == order
    # undecided = every pair the 16-bit aligner refused, whether the planner saw it coming (empty target) or the DP overflowed
    # on the GPU (the 40 kb target): the caller's linear-memory fallback decides those (searchcore.cpp:806-832)
    und = sorted(int(k) for k in range(len(qi)) if full.verdict[k] == 0)
    assert sorted(rk["undecided"].tolist()) == und and len(und) == (n - 1) + (n - 2)
    overflowed = [k for k in und if int(ti[k]) == n - 2]
    assert len(overflowed) == n - 2 and all(full.row(k)[0] == 32767 and full.row(k)[5] == "" for k in overflowed)
