"""-m gpu: the drop-in proof.  oracle/_ref/vsearch_vsx is the REFERENCE CLI (every translation unit of torognes/vsearch
compiled in place by oracle/Makefile) with exactly one unit swapped: src/core/align_simd.cpp is replaced by
shim/vsx_search16_shim.cpp, which defines the same four symbols (search16_init/exit/qprep/search16) on top of libvsx.so.
Every command that reaches the aligner must write byte-identical output files to the unmodified CLI
(oracle/_ref/vsearch_ref): same hits, %id, CIGARs, alignments, clusters, consensus sequences."""
import os
import random
import subprocess

import pytest

from tests import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vsearch_ref")
VSX_BIN = os.path.join(ROOT, "oracle", "_ref", "vsearch_vsx")
FIELDS = "query+target+id+alnlen+mism+opens+exts+raw+caln+qlo+qhi+tlo+thi+id0+id1+id2+id3+id4+qrow+trow"


def _fasta(path, seqs, prefix):
    with open(path, "w") as f:
        f.write("".join(f">{prefix}{i};size={1 + (i * 7) % 5}\n{s}\n" for i, s in enumerate(seqs)))


def _run_both(tmp, args_of):
    """args_of(outdir) -> (argv tail, [output file names]); runs both binaries, returns {name: (ref bytes, vsx bytes)}"""
    res = {}
    for tag, exe in (("ref", REF_BIN), ("vsx", VSX_BIN)):
        out = os.path.join(tmp, tag)
        os.makedirs(out, exist_ok=True)
        argv, files = args_of(out)
        p = subprocess.run([exe] + argv + ["--quiet"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        for name in files:
            with open(os.path.join(out, name), "rb") as f:
                data = f.read()
            # --alnout / --uchimealns echo the command line: neutralise the binary name and the output directory
            data = data.replace(exe.encode(), b"VSEARCH").replace(out.encode(), b"OUT")
            res.setdefault(name, {})[tag] = data
    return res


def _check(res):
    for name, d in res.items():
        assert len(d["ref"]) > 0, name
        if d["ref"] != d["vsx"]:
            a, b = d["ref"].splitlines(), d["vsx"].splitlines()
            for i, (x, y) in enumerate(zip(a, b)):
                assert x == y, f"{name} line {i}:\n ref {x[:300]}\n vsx {y[:300]}"
            assert len(a) == len(b), f"{name}: {len(a)} vs {len(b)} lines"


@pytest.fixture
def binaries(gpu_required):
    for b in (REF_BIN, VSX_BIN):
        if not os.path.exists(b):
            pytest.fail(f"{b} missing: run `make -C oracle ref_full ref_shim` in the build container (it travels with the repo)")


def _inputs(seed):
    rng = random.Random(seed)
    db, _ = common.family_db(rng, 12, 8, 380, div=0.07)
    db += [common.rnd_seq(rng, rng.randint(150, 450)) for _ in range(15)]
    db += [common.mutate(rng, db[5], 0.03, "ACGTN"), common.mutate(rng, db[9], 0.03, "ACGTRYKM")]
    qs, _ = common.queries_from_db(rng, db, 60, 170)
    qs += [common.mutate(rng, db[rng.randrange(len(db))], 0.06) for _ in range(15)]
    qs += [common.rnd_seq(rng, 90), "ACGT" * 25]
    return db, qs


@pytest.mark.parametrize("extra", [
    ["--id", "0.9"],
    ["--id", "0.7", "--maxaccepts", "3", "--maxrejects", "16", "--strand", "both"],
    ["--id", "0.8", "--maxaccepts", "0", "--maxrejects", "0", "--gapopen", "10I/3E", "--gapext", "1I/1E", "--mismatch", "-2"],
], ids=["id90", "both_strands", "custom_scoring_all_hits"])
def test_usearch_global_through_the_shim(binaries, tmp_path, extra):
    db, qs = _inputs(41)
    _fasta(tmp_path / "db.fa", db, "t")
    _fasta(tmp_path / "q.fa", qs, "q")

    def args(out):
        return (["--usearch_global", str(tmp_path / "q.fa"), "--db", str(tmp_path / "db.fa"), "--qmask", "none", "--dbmask", "none",
                 "--threads", "1", "--userout", os.path.join(out, "u.tsv"), "--userfields", FIELDS,
                 "--alnout", os.path.join(out, "aln.txt"), "--uc", os.path.join(out, "hits.uc"),
                 "--samout", os.path.join(out, "hits.sam")] + extra, ["u.tsv", "aln.txt", "hits.uc", "hits.sam"])
    _check(_run_both(str(tmp_path), args))


def test_cluster_fast_through_the_shim(binaries, tmp_path):
    rng = random.Random(8)
    seqs, _ = common.family_db(rng, 10, 9, 300, div=0.04)
    seqs += [common.rnd_seq(rng, rng.randint(200, 350)) for _ in range(10)]
    rng.shuffle(seqs)
    _fasta(tmp_path / "in.fa", seqs, "s")

    def args(out):
        return (["--cluster_fast", str(tmp_path / "in.fa"), "--id", "0.9", "--qmask", "none", "--threads", "1", "--sizein", "--sizeout",
                 "--uc", os.path.join(out, "c.uc"), "--centroids", os.path.join(out, "cent.fa"),
                 "--msaout", os.path.join(out, "msa.fa"), "--consout", os.path.join(out, "cons.fa"),
                 "--profile", os.path.join(out, "prof.txt")], ["c.uc", "cent.fa", "msa.fa", "cons.fa", "prof.txt"])
    _check(_run_both(str(tmp_path), args))


def test_allpairs_global_through_the_shim(binaries, tmp_path):
    rng = random.Random(13)
    seqs, _ = common.family_db(rng, 5, 7, 260, div=0.08)
    _fasta(tmp_path / "in.fa", seqs, "s")

    def args(out):
        return (["--allpairs_global", str(tmp_path / "in.fa"), "--id", "0.75", "--qmask", "none", "--threads", "1",
                 "--userout", os.path.join(out, "u.tsv"), "--userfields", FIELDS, "--alnout", os.path.join(out, "aln.txt")],
                ["u.tsv", "aln.txt"])
    _check(_run_both(str(tmp_path), args))


def test_uchime_ref_through_the_shim(binaries, tmp_path):
    """chimera detection aligns query segments against parents with search16 (chimera.cpp:1899-2078)"""
    rng = random.Random(21)
    parents = [common.rnd_seq(rng, 400) for _ in range(6)]
    qs = []
    for _ in range(12):
        a, b = rng.sample(range(6), 2)
        cut = rng.randint(120, 280)
        qs.append(common.mutate(rng, parents[a][:cut] + parents[b][cut:], 0.01))
    qs += [common.mutate(rng, parents[k], 0.02) for k in range(6)]
    _fasta(tmp_path / "ref.fa", parents, "p")
    _fasta(tmp_path / "q.fa", qs, "q")

    def args(out):
        return (["--uchime_ref", str(tmp_path / "q.fa"), "--db", str(tmp_path / "ref.fa"), "--qmask", "none", "--dbmask", "none",
                 "--threads", "1", "--uchimeout", os.path.join(out, "uchime.tsv"), "--uchimealns", os.path.join(out, "alns.txt"),
                 "--chimeras", os.path.join(out, "chim.fa"), "--nonchimeras", os.path.join(out, "non.fa")],
                ["uchime.tsv", "alns.txt", "chim.fa", "non.fa"])
    _check(_run_both(str(tmp_path), args))


def test_four_reference_threads_share_the_gpu(binaries, tmp_path):
    """--threads 4: four s16info_s contexts (one per reference worker thread, searchcore.hpp:151) drive the same GPU at once;
    the hits are the single-threaded CLI's (line order depends on thread timing, so lines are compared sorted)"""
    db, qs = _inputs(97)
    _fasta(tmp_path / "db.fa", db, "t")
    _fasta(tmp_path / "q.fa", qs * 3, "q")
    out = {}
    for tag, exe, threads in (("ref", REF_BIN, "1"), ("vsx", VSX_BIN, "4")):
        uf = str(tmp_path / f"{tag}.tsv")
        p = subprocess.run([exe, "--usearch_global", str(tmp_path / "q.fa"), "--db", str(tmp_path / "db.fa"), "--qmask", "none",
                            "--dbmask", "none", "--threads", threads, "--id", "0.8", "--maxaccepts", "2", "--userout", uf,
                            "--userfields", FIELDS, "--quiet"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        out[tag] = sorted(open(uf).read().splitlines())
    assert len(out["ref"]) > 100
    assert out["ref"] == out["vsx"]


# ---- the reference's LIBRARY API on the fast path (shim/vsx_api_adapter.cpp; oracle/Makefile ref_api) --------------------------
API_SEARCH = os.path.join(ROOT, "oracle", "_ref", "example_search_vsx")
API_CLUSTER = os.path.join(ROOT, "oracle", "_ref", "example_cluster_vsx")
API_DRIVER = os.path.join(ROOT, "oracle", "_ref", "api_driver_vsx")


@pytest.fixture
def api_binaries(gpu_required):
    for b in (API_SEARCH, API_CLUSTER, API_DRIVER):
        if not os.path.exists(b):
            pytest.fail(f"{b} missing: run `make -C oracle ref_full ref_api` in the build container (it travels with the repo)")


def _api_data(tmp):
    """api_examples/data as the reference's examples expect it in their working directory, from tests/golden/ref_api_examples.json"""
    ex = common.load_api_examples()
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    with open(os.path.join(tmp, "data", "chimera_ref.fasta"), "w") as f:
        f.write("".join(f">{n}\n{s}\n" for n, s in ex["refs"].items()))
    with open(os.path.join(tmp, "data", "chimera_queries.fasta"), "w") as f:
        f.write("".join(f">{n}\n{s}\n" for n, s in ex["queries"].items()))
    return ex


def _api_run(argv, cwd, **extra_env):
    env = dict(os.environ, VSX_ADAPTER_TRACE="1", **extra_env)
    return subprocess.run(argv, capture_output=True, text=True, timeout=600, cwd=cwd, env=env)


def _assert_fast(p, marker):
    """the adapter prints `marker` only AFTER the fast path has answered; a runtime fallback to the renamed reference functions
    (which would make every 'equals the sequential reference' comparison vacuous) prints one of the other two"""
    assert marker in p.stderr, p.stderr[-2000:]
    assert "fast path failed" not in p.stderr, p.stderr[-2000:]
    assert "reference code" not in p.stderr, p.stderr[-2000:]


def test_reference_example_search_on_the_fast_path(api_binaries, tmp_path):
    """the reference's own api_examples/example_search.cc, search_batch bound to the GPU path: its Part 2 asserts
    search_batch == search_session_single (the reference's code) field by field; Part 1's TSV is the in-tree golden file"""
    ex = _api_data(str(tmp_path))
    p = _api_run([API_SEARCH], str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    _assert_fast(p, "search_batch -> vsx_multi_search_batch")
    assert "PASS: batch search matches sequential search" in p.stderr
    got = sorted(tuple(l.split("\t")) for l in p.stdout.splitlines())
    exp = sorted((e["query"], e["target"], e["id"]) for e in ex["expected_search"])
    assert got == exp


def test_reference_example_cluster_on_the_fast_path(api_binaries, tmp_path):
    """api_examples/example_cluster.cc: cluster_assign_batch (GPU path) == cluster_assign_single (reference code), and the H
    records of the in-tree expected_cluster.uc"""
    ex = _api_data(str(tmp_path))
    p = _api_run([API_CLUSTER], str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    _assert_fast(p, "cluster_assign_batch -> vsx_cluster_fast")
    assert "PASS: batch cluster matches sequential" in p.stderr
    hits = sorted(l.split("\t") for l in p.stdout.splitlines() if l.startswith("H"))
    exp = sorted(ex["expected_cluster_hits"], key=lambda e: e["query"])
    assert [(h[8], h[9], h[3], h[7]) for h in sorted(hits, key=lambda h: h[8])] == [(e["query"], e["target"], e["id"], e["cigar"]) for e in exp]


@pytest.mark.parametrize("strand,qmask,dbmask", [(0, "dust", "dust"), (1, "dust", "dust"), (1, "none", "none"), (0, "soft", "soft"), (0, "dust", "none"),
                                                  # r06: --hardmask on the fast path (the Database arrives hard-masked, the queries are masked by libvsx)
                                                  (1, "soft+hard", "soft+hard"), (1, "dust+hard", "dust+hard")])
def test_library_search_batch_equals_sequential_reference(api_binaries, tmp_path, strand, qmask, dbmask):
    """a larger embedder run (oracle/api_driver.cc): every field of every search_result_s of search_batch (GPU path) against the
    reference's sequential search_session_single, with gapped alignments, both strands and the reference's default DUST masking"""
    from tests import test_gpu_mask as M
    hard = qmask.endswith("+hard")
    qmask, dbmask = qmask.split("+")[0], dbmask.split("+")[0]
    rng = random.Random(31 + strand)
    db = M._masked_families(rng, 60, 5, 420, 0.05, qmask == "soft" or hard)
    qs = M._queries(rng, db, 400, 220, 0.04, qmask == "soft" or hard)
    if strand:
        for k in range(0, len(qs), 2):
            qs[k] = "".join(M.COMP[c] for c in reversed(qs[k]))
    M._write(str(tmp_path / "db.fa"), [f"t{i}" for i in range(len(db))], db)
    M._write(str(tmp_path / "q.fa"), [f"q{i}" for i in range(len(qs))], qs)
    p = _api_run([API_DRIVER, "search", str(tmp_path / "db.fa"), str(tmp_path / "q.fa"), "0.7" if hard else "0.8", "3", "8", str(strand), qmask, dbmask] + (["hardmask=1"] if hard else []),
                 str(tmp_path))
    assert p.returncode == 0, (p.stdout, p.stderr[-3000:])
    _assert_fast(p, "search_batch -> vsx_multi_search_batch")
    n_hits = int(p.stdout.split(" queries, ")[1].split(" hits")[0])
    assert n_hits > (300 if hard else 500) and p.stdout.strip().endswith(" 0 differences"), p.stdout
    if strand:
        assert int(p.stdout.split("(")[1].split(" on the minus")[0]) > 100


@pytest.mark.parametrize("batch,qmask", [(64, "dust"), (100000, "dust"), (17, "none")])
def test_library_cluster_batch_equals_sequential_reference(api_binaries, tmp_path, batch, qmask):
    """cluster_assign_batch in ranges of `batch` (GPU path) against cluster_assign_single (reference code): every field of every
    cluster_result_s, CIGAR strings included"""
    from tests import test_gpu_mask as M
    rng = random.Random(5)
    seqs = M._masked_families(rng, 40, 8, 300, 0.03, False)
    rng.shuffle(seqs)
    M._write(str(tmp_path / "c.fa"), [f"s{i:04d}" for i in range(len(seqs))], seqs)
    p = _api_run([API_DRIVER, "cluster", str(tmp_path / "c.fa"), "0.9", "1", "8", str(batch), qmask], str(tmp_path))
    assert p.returncode == 0, (p.stdout, p.stderr[-3000:])
    _assert_fast(p, "cluster_assign_batch -> vsx_cluster_fast")
    assert p.stdout.strip().endswith(" 0 differences"), p.stdout
    assert int(p.stdout.split(" clusters, ")[1].split(" members")[0]) > 100


def test_library_search_batch_on_two_replicas(api_binaries, tmp_path):
    """VSX_DEVICES=0,0: the adapter's searcher is a vsx_multi_searcher with two database replicas (here on one GPU); the queries
    are sharded over them and the merged result must still equal the reference's sequential search field by field"""
    from tests import test_gpu_mask as M
    rng = random.Random(77)
    db = M._masked_families(rng, 50, 5, 400, 0.05, False)
    qs = M._queries(rng, db, 301, 200, 0.04, False)
    M._write(str(tmp_path / "db.fa"), [f"t{i}" for i in range(len(db))], db)
    M._write(str(tmp_path / "q.fa"), [f"q{i}" for i in range(len(qs))], qs)
    p = _api_run([API_DRIVER, "search", str(tmp_path / "db.fa"), str(tmp_path / "q.fa"), "0.8", "3", "8", "1", "dust", "dust"], str(tmp_path),
                 VSX_DEVICES="0,0")
    assert p.returncode == 0, (p.stdout, p.stderr[-3000:])
    _assert_fast(p, "search_batch -> vsx_multi_search_batch")
    assert int(p.stdout.split(" queries, ")[1].split(" hits")[0]) > 300 and p.stdout.strip().endswith(" 0 differences"), p.stdout


def test_library_zero_accepts_examines_no_candidate(api_binaries, tmp_path):
    """maxaccepts == 0 is not 'unlimited' in the library (search.cpp:521-529 does not rewrite it): the reference's candidate loop
    never runs (searchcore.cpp:915-918).  The adapter answers that directly; the driver compares with the reference's sequential
    search, which must also report nothing"""
    from tests import test_gpu_mask as M
    rng = random.Random(78)
    db = M._masked_families(rng, 10, 4, 300, 0.05, False)
    qs = M._queries(rng, db, 40, 200, 0.04, False)
    M._write(str(tmp_path / "db.fa"), [f"t{i}" for i in range(len(db))], db)
    M._write(str(tmp_path / "q.fa"), [f"q{i}" for i in range(len(qs))], qs)
    for ma, mr in (("0", "8"), ("2", "0")):
        p = _api_run([API_DRIVER, "search", str(tmp_path / "db.fa"), str(tmp_path / "q.fa"), "0.8", ma, mr, "0", "none", "none"], str(tmp_path))
        assert p.returncode == 0, (p.stdout, p.stderr[-3000:])
        assert "no candidates are examined" in p.stderr
        assert " 0 hits" in p.stdout and p.stdout.strip().endswith(" 0 differences"), p.stdout
