"""-m gpu: parity AT SCALE against the reference CLI (oracle/_ref/vsearch_ref) for the BASELINE shapes that the small
byte-parity tests do not reach:

  config 3  --cluster_fast   50 000 x 300 bp amplicons at 2 % divergence, --id 0.97: 13 rounds of 4 096 with centroid-index
            rebuilds between them and the intra-round fix-up (core/cluster.cpp:877-1125, :601-856)
  config 4  --allpairs_global 2 000 x 400 bp at --id 0.8: 2.0 M pairs through the pipelined pair-list slices and the device
            accept filter (commands/allpairs_global.cpp:394-527)
  config 5  --usearch_global 150 bp queries vs a 262 144-sequence DB: 8 tiles of the device k-mer index with real candidate
            competition (core/searchcore.cpp:260-340)

Every comparison is file against file (sorted where the reference's own thread scheduling decides the line order)."""
import os
import subprocess
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vsearch_ref")
FIELDS = ["query", "target", "id", "alnlen", "mism", "opens", "exts", "raw", "caln", "id0", "id1", "id2", "id3", "id4"]


def _need_ref():
    if not os.path.exists(REF_BIN):
        pytest.fail("oracle/_ref/vsearch_ref missing: run `make -C oracle ref_full` in the build container")


def _threads():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def _strings(flat, off, ln):
    b = flat.cpu().numpy().tobytes()
    return [b[int(o):int(o) + int(l)].decode() for o, l in zip(off, ln)]


def _write(path, names, seqs):
    with open(path, "w") as f:
        f.write("".join(f">{n}\n{s}\n" for n, s in zip(names, seqs)))


def _first_diff(got, exp):
    for i, (a, b) in enumerate(zip(got, exp)):
        if a != b:
            return f"line {i}:\n got {a}\n exp {b}"
    return f"length {len(got)} vs {len(exp)}"


def _run(cmd):
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return time.time() - t0


def test_cluster_fast_50k_matches_reference_cli(gpu_required, tmp_path):
    _need_ref()
    from vsearch_amd import Aligner, SearchSession, workload
    n = 50_000
    flat, off, ln, fam = workload.make_family_db(n, 300, members=50, div=0.02, seed=23, device="cpu")
    seqs = _strings(flat, off, ln)
    rng = np.random.default_rng(1)
    perm = rng.permutation(n)                               # input order: shuffled (the command sorts by length itself)
    seqs = [seqs[i] for i in perm]
    names = [f"s{i:06d}" for i in range(n)]
    tmp = str(tmp_path)
    _write(tmp + "/c.fa", names, seqs)
    t_ref = _run([REF_BIN, "--cluster_fast", tmp + "/c.fa", "--id", "0.97", "--qmask", "none", "--threads", str(_threads()),
                  "--uc", tmp + "/c.uc", "--quiet"])
    exp = open(tmp + "/c.uc").read().splitlines()
    # Database::sortbylength (core/db.cpp:433-450): length descending, then label
    order = sorted(range(n), key=lambda i: (-len(seqs[i]), names[i]))
    sseqs, snames = [seqs[i] for i in order], [names[i] for i in order]
    with Aligner() as al:
        ss = SearchSession(al, sseqs, id=0.97, maxrejects=8)          # --cluster_fast default maxrejects (cli.cc:4163-4167)
        t0 = time.time()
        got = ss.uc_lines(snames, round=4096)
        t_vsx = time.time() - t0
        stats = dict(ss.stats)
    nh = sum(1 for l in exp if l[0] == "H")
    assert nh > 25_000 and sum(1 for l in exp if l[0] == "S") >= 1000
    assert got == exp, _first_diff(got, exp)
    assert stats["stages"] >= 12                                      # 13 rounds; the first one meets an empty centroid index
    print(f"cluster_fast 50k: reference {t_ref:.1f} s ({_threads()} threads), vsx {t_vsx:.1f} s, {stats['pairs_aligned']} pairs")


def test_allpairs_2000_matches_reference_cli(gpu_required, tmp_path):
    _need_ref()
    from vsearch_amd import Aligner, SearchSession, workload
    n = 2000
    flat, off, ln, fam = workload.make_family_db(n, 400, members=50, div=0.10, seed=29, device="cpu")
    seqs = _strings(flat, off, ln)
    names = [f"t{i}" for i in range(n)]
    tmp = str(tmp_path)
    _write(tmp + "/a.fa", names, seqs)
    t_ref = _run([REF_BIN, "--allpairs_global", tmp + "/a.fa", "--id", "0.8", "--qmask", "none", "--threads", str(_threads()),
                  "--userout", tmp + "/ua.tsv", "--userfields", "+".join(FIELDS), "--quiet"])
    exp = sorted(open(tmp + "/ua.tsv").read().splitlines())
    with Aligner() as al:
        ss = SearchSession(al, seqs, id=0.8)
        t0 = time.time()
        hits = ss.allpairs(0, n)                                       # one block: 1 999 000 pairs = pipelined slices
        t_vsx = time.time() - t0
        pairs = ss.stats["pairs_aligned"]
        got = sorted(ss.userout(seqs, qnames=names, tnames=names, fields=FIELDS, hits=hits))
    assert pairs == n * (n - 1) // 2
    assert len(exp) > 20_000
    assert got == exp, _first_diff(got, exp)
    print(f"allpairs 2000 x 400: reference {t_ref:.1f} s ({_threads()} threads), vsx {t_vsx:.1f} s, {len(exp)} accepted pairs")


def test_search_150bp_vs_262144_db_matches_reference_cli(gpu_required, tmp_path):
    _need_ref()
    from vsearch_amd import Aligner, SearchSession, workload
    n_db, n_q = 262_144, 4000
    flat, off, ln, fam = workload.make_family_db(n_db, 300, members=64, div=0.08, seed=31, device="cpu")
    qflat, qoff, qln, src = workload.make_queries(flat, off, ln, n_q, 150, seed=37, device="cpu")
    db, qs = _strings(flat, off, ln), _strings(qflat, qoff, qln)
    tn, qn = [f"t{i}" for i in range(n_db)], [f"q{i}" for i in range(n_q)]
    tmp = str(tmp_path)
    _write(tmp + "/db.fa", tn, db)
    _write(tmp + "/q.fa", qn, qs)
    extra = ["--id", "0.9", "--maxaccepts", "2"]
    t_ref = _run([REF_BIN, "--usearch_global", tmp + "/q.fa", "--db", tmp + "/db.fa", "--qmask", "none", "--dbmask", "none",
                  "--threads", str(_threads()), "--userout", tmp + "/u.tsv", "--userfields", "+".join(FIELDS), "--quiet"] + extra)
    exp = sorted(open(tmp + "/u.tsv").read().splitlines())
    with Aligner() as al:
        ss = SearchSession(al, db, id=0.9, maxaccepts=2)
        t0 = time.time()
        hits = ss.search_batch(qs)
        t_vsx = time.time() - t0
        got = sorted(ss.userout(qs, qnames=qn, tnames=tn, fields=FIELDS, hits=hits))
        stats = dict(ss.stats)
    assert len(exp) > 3000
    assert got == exp, _first_diff(got, exp)
    print(f"search 4000 x 150 bp vs 262144: reference {t_ref:.1f} s, vsx {t_vsx:.1f} s, {stats['pairs_aligned']} pairs")
