"""-m gpu: masking before the k-mer stage (--qmask / --dbmask), on the device k-mer path.

  soft_mask=1  lower-case symbols are left out of the k-mers            core/unique.cpp:198-199, utils/maps.cpp (map_mask_lower)
  soft_mask=2  DUST (the reference commands' DEFAULT): database masked by vsx_mask.hip, queries by vsx_mask.cpp
               core/mask.cpp:79-199, commands/usearch_global.cpp:386-392,577-583, core/search.cpp:294-303

Each against the reference CLI itself (oracle/_ref/vsearch_ref) run with the same masking options -- for dust that is the
reference's plain default command line, no --qmask / --dbmask at all.
"""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vsearch_ref")
FIELDS = ["query", "target", "id", "alnlen", "mism", "opens", "exts", "raw", "caln", "qstrand"]
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a", "N": "N", "R": "Y", "Y": "R"}


def _need_ref():
    if not os.path.exists(REF_BIN):
        pytest.fail("oracle/_ref/vsearch_ref missing: run `make -C oracle ref_full` in the build container")


def _first_diff(got, exp):
    for i, (a, b) in enumerate(zip(got, exp)):
        if a != b:
            return f"line {i}:\n got {a}\n exp {b}"
    return f"length {len(got)} vs {len(exp)}"


def _write(path, names, seqs):
    with open(path, "w") as f:
        f.write("".join(f">{n}\n{s}\n" for n, s in zip(names, seqs)))


def _low_complexity(rng, n):
    p = rng.choice([1, 2, 3, 4])
    u = "".join(rng.choice("ACGT") for _ in range(p))
    return (u * (n // p + 1))[:n]


def _masked_families(rng, n_families, members, length, div, lower_case):
    """families whose ancestors carry low-complexity stretches (shared ACROSS families, so unmasked they drag unrelated
    targets into the candidate lists) and, with lower_case, soft-masked stretches of ordinary sequence"""
    shared = [_low_complexity(rng, rng.randint(30, 70)) for _ in range(3)]
    db = []
    for _ in range(n_families):
        parts, left = [], length
        while left > 0:
            if rng.random() < 0.35:
                seg = rng.choice(shared)
            else:
                seg = common.rnd_seq(rng, rng.randint(20, 90))
                if lower_case and rng.random() < 0.3:
                    seg = seg.lower()
            parts.append(seg)
            left -= len(seg)
        anc = "".join(parts)
        for _ in range(members):
            m = common.mutate(rng, anc.upper(), div)
            # keep the ancestor's case pattern position by position where the lengths still agree (soft masking is about case)
            if lower_case:
                m = "".join(c.lower() if k < len(anc) and anc[k].islower() else c for k, c in enumerate(m))
            db.append(m)
    return db


def _queries(rng, db, n, qlen, div, lower_case):
    qs = []
    for _ in range(n):
        s = db[rng.randrange(len(db))]
        a = rng.randrange(max(1, len(s) - qlen))
        frag = s[a:a + qlen]
        m = common.mutate(rng, frag.upper(), div)
        if lower_case:
            m = "".join(c.lower() if k < len(frag) and frag[k].islower() else c for k, c in enumerate(m))
        qs.append(m)
    return qs


def _reference_userout(tmp, db, qs, mask_args, extra):
    _write(tmp + "/db.fa", [f"t{i}" for i in range(len(db))], db)
    _write(tmp + "/q.fa", [f"q{i}" for i in range(len(qs))], qs)
    p = subprocess.run([REF_BIN, "--usearch_global", tmp + "/q.fa", "--db", tmp + "/db.fa", "--threads", "1", "--userout", tmp + "/u.tsv",
                        "--userfields", "+".join(FIELDS), "--quiet"] + mask_args + extra, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return open(tmp + "/u.tsv").read().splitlines()


@pytest.mark.parametrize("mode,mask_args,lower_case", [
    (1, ["--qmask", "soft", "--dbmask", "soft"], True),
    (2, [], False),                                             # the reference's default command line: dust on both sides
    (2, [], True),                                              # dust upper-cases its input first: the input's case must not matter
])
@pytest.mark.parametrize("strand_both", [0, 1])
def test_masked_search_matches_reference_cli(gpu_required, tmp_path, mode, mask_args, lower_case, strand_both):
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(100 * mode + 10 * lower_case + strand_both)
    db = _masked_families(rng, 40, 5, 420, 0.05, lower_case)
    qs = _queries(rng, db, 150, 200, 0.03, lower_case)
    if strand_both:
        for k in range(0, len(qs), 2):
            qs[k] = "".join(COMP[c] for c in reversed(qs[k]))
    extra = ["--id", "0.8", "--maxaccepts", "2", "--maxrejects", "4"] + (["--strand", "both"] if strand_both else [])
    tmp = str(tmp_path)
    exp = _reference_userout(tmp, db, qs, mask_args, extra)
    unmasked = _reference_userout(tmp, db, qs, ["--qmask", "none", "--dbmask", "none"], extra)
    with Aligner() as al:
        ss = SearchSession(al, db, id=0.8, maxaccepts=2, maxrejects=4, strand_both=strand_both, soft_mask=mode)
        got = ss.userout(qs, fields=FIELDS)
        # candidate lists: device counters (refused with EINVAL if masking pushed the searcher off the device k-mer path) == host
        # restatement, under the same masking
        sub = qs[:40]
        dev, host = ss.candidates_batch(sub, device=True), ss.candidates_batch(sub, device=False)
    assert len(exp) > 100
    assert got == exp, _first_diff(got, exp)
    assert exp != unmasked, "the data set does not exercise the masking"
    assert dev == host


@pytest.mark.parametrize("mode,mask_args", [
    (1, ["--qmask", "soft", "--dbmask", "soft"]),
    (2, []),                                                    # dust on both sides (the default) + --hardmask
])
@pytest.mark.parametrize("strand_both", [0, 1])
def test_hardmask_search_matches_reference_cli(gpu_required, tmp_path, mode, mask_args, strand_both):
    """r06, --hardmask (core/mask.cpp:137-191,248-271; core/search.cpp:294-303): masked symbols become 'N' in the sequence, so the k-mer
    stage AND the alignment see them -- scores, %id and CIGARs change against the soft-masked run.  vsx_search_opts::hardmask = 3 against the
    reference CLI's own --hardmask run: soft masking (every lower-case symbol) and DUST (the intervals, input case kept), both strands."""
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(700 + 10 * mode + strand_both)
    db = _masked_families(rng, 40, 5, 420, 0.05, True)
    qs = _queries(rng, db, 150, 200, 0.03, True)
    if strand_both:
        for k in range(0, len(qs), 2):
            qs[k] = "".join(COMP[c] for c in reversed(qs[k]))
    extra = ["--id", "0.7", "--maxaccepts", "2", "--maxrejects", "4"] + (["--strand", "both"] if strand_both else [])
    tmp = str(tmp_path)
    exp = _reference_userout(tmp, db, qs, mask_args + ["--hardmask"], extra)
    soft = _reference_userout(tmp, db, qs, mask_args, extra)
    with Aligner() as al:
        ss = SearchSession(al, db, id=0.7, maxaccepts=2, maxrejects=4, strand_both=strand_both, soft_mask=mode, hardmask=3)
        got = ss.userout(qs, fields=FIELDS)
        sub = qs[:40]
        dev, host = ss.candidates_batch(sub, device=True), ss.candidates_batch(sub, device=False)
        # a caller whose database text is hard-masked already (the library API): hardmask = 2 on that text gives the same answer
        ss2 = SearchSession(al, ss.masked_db_text(), id=0.7, maxaccepts=2, maxrejects=4, strand_both=strand_both, soft_mask=1, qmask=1 + mode, hardmask=2)
        got2 = ss2.userout(qs, fields=FIELDS)
    assert len(exp) > 80
    assert got == exp, _first_diff(got, exp)
    assert exp != soft, "the data set does not exercise --hardmask"
    assert dev == host
    assert got2 == exp, _first_diff(got2, exp)


def test_hardmask_cluster_fast_matches_reference_cli(gpu_required, tmp_path):
    """--cluster_fast --hardmask (cluster.cpp:1192-1197: dust with 'N', or hardmask_all under --qmask soft)"""
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    for mode, mask in ((2, []), (1, ["--qmask", "soft"])):
        rng = random.Random(4200 + mode)
        seqs = _masked_families(rng, 30, 7, 300, 0.02, True)
        rng.shuffle(seqs)
        names = [f"s{i:04d}" for i in range(len(seqs))]
        f_in, f_uc = str(tmp_path / "c.fa"), str(tmp_path / "c.uc")
        _write(f_in, names, seqs)
        outs = {}
        for key, more in (("hard", ["--hardmask"]), ("plain", [])):
            p = subprocess.run([REF_BIN, "--cluster_fast", f_in, "--id", "0.9", "--threads", "1", "--uc", f_uc, "--quiet", "--maxaccepts", "1",
                                "--maxrejects", "2"] + mask + more, capture_output=True, text=True)
            assert p.returncode == 0, p.stderr
            outs[key] = open(f_uc).read().splitlines()
        order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), names[i]))
        sseqs, snames = [seqs[i] for i in order], [names[i] for i in order]
        with Aligner() as al:
            ss = SearchSession(al, sseqs, id=0.9, maxaccepts=1, maxrejects=2, soft_mask=mode, hardmask=3)
            got = ss.uc_lines(snames, round=32)
        assert got == outs["hard"], (mode, _first_diff(got, outs["hard"]))
        assert outs["hard"] != outs["plain"], "the data set does not exercise --hardmask"


def test_device_dust_bits_match_host_dust(gpu_required):
    """vsx_mask.hip (one wave per sequence) against vsx_mask.cpp, which tests/test_host_cpu.py pins to the reference CLI:
    the same intervals, bit for bit, on the golden inputs and on a larger random set with window-boundary lengths"""
    import json
    from vsearch_amd import Aligner, dust_mask, _lib
    from oracle import gen_golden
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "dust_golden.json")))
    seqs = list(doc["in"]) + gen_golden.dust_inputs(random.Random(77), 3000) + ["A" * 70000, "ACGT" * 3, ""]
    lib = _lib.load()
    blob, off, lens = common.blobify(seqs)
    nb = (len(blob) + 7) // 8
    with Aligner() as al:
        for mode in (2, 1):
            bits = np.zeros(nb, np.uint8)
            rc = lib.vsx_internal_mask_bits(al.h, C.c_uint64(len(lens)), C.cast(C.c_char_p(blob), C.c_void_p), C.c_uint64(len(blob)),
                                            off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), C.c_int(mode),
                                            bits.ctypes.data_as(C.c_void_p))
            assert rc == 0, lib.vsx_last_error()
            got = np.unpackbits(bits, bitorder="little")[:len(blob)].astype(bool)
            text = b"".join(dust_mask(seqs)) if mode == 2 else blob
            arr = np.frombuffer(text, np.uint8)
            exp = ~np.isin(arr, np.frombuffer(b"ACGTU", np.uint8))
            bad = np.nonzero(got != exp)[0]
            assert bad.size == 0, (mode, int(bad[0]), int(bad.size))
            if mode == 2:
                assert exp.sum() > 10000


def test_dust_cluster_fast_matches_reference_cli_defaults(gpu_required, tmp_path):
    """--cluster_fast with the reference's default masking (dust_all before clustering, cluster.cpp:1192)"""
    _need_ref()
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(42)
    seqs = _masked_families(rng, 30, 7, 300, 0.02, False)
    rng.shuffle(seqs)
    names = [f"s{i:04d}" for i in range(len(seqs))]
    f_in, f_uc = str(tmp_path / "c.fa"), str(tmp_path / "c.uc")
    _write(f_in, names, seqs)
    outs = {}
    for key, mask in (("dust", []), ("none", ["--qmask", "none"])):
        p = subprocess.run([REF_BIN, "--cluster_fast", f_in, "--id", "0.95", "--threads", "1", "--uc", f_uc, "--quiet", "--maxaccepts", "1",
                            "--maxrejects", "2"] + mask, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        outs[key] = open(f_uc).read().splitlines()
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), names[i]))
    sseqs, snames = [seqs[i] for i in order], [names[i] for i in order]
    with Aligner() as al:
        ss = SearchSession(al, sseqs, id=0.95, maxaccepts=1, maxrejects=2, soft_mask=2)
        got = ss.uc_lines(snames, round=32)
    assert got == outs["dust"], _first_diff(got, outs["dust"])
    assert sum(1 for l in got if l[0] == "H") > 50
