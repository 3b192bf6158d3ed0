#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: randomized soak of the dispatch layer (vsx_search_batch: device k-mer stage, accept / reject replay,
filters, ranking, both strands, masking) against the REFERENCE CLI itself (oracle/_ref/vsearch_ref --usearch_global).

Every round draws an option set -- identity definition and thresholds, maxaccepts / maxrejects, word length, strand, masking on
either side, the unaligned and aligned filters, abundance filters with --sizein, --self / --selfid, a scoring set expressed in the
CLI's own --match / --mismatch / --gapopen / --gapext syntax -- and a small family-structured data set with low-complexity and
lower-case stretches, and compares the --userout lines (15 fields incl. CIGAR, the five identities and the strand) byte for byte.

    python oracle/soak_search.py --seconds 120 --seed 1 --out gpurun_out/soak_search.json
"""
import argparse
import json
import os
import random
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import refcli  # noqa: E402
from tests import common  # noqa: E402
from tests import test_gpu_mask as M  # noqa: E402

FIELDS = ["query", "target", "id", "alnlen", "mism", "opens", "exts", "raw", "caln", "id0", "id1", "id2", "id3", "id4", "qstrand"]
MASKS = ["none", "soft", "dust"]


WORDLENGTHS = None            # --wordlengths LO..HI: every round draws its word length from this range


def draw_options(rng):
    """-> (SearchSession kwargs, scoring tuple or None, CLI argv)"""
    o, cli = {}, []

    def put(key, val, flag, text=None):
        o[key] = val
        cli.extend([flag, text if text is not None else repr(val) if isinstance(val, float) else str(val)])

    put("id", rng.choice([0.5, 0.7, 0.8, 0.9, 0.95, 0.97]), "--id")
    put("maxaccepts", rng.choice([0, 1, 1, 2, 3, 5]), "--maxaccepts")
    put("maxrejects", rng.choice([0, 2, 8, 16, 32]), "--maxrejects")
    if WORDLENGTHS is not None:
        put("wordlength", rng.randint(*WORDLENGTHS), "--wordlength")
    elif rng.random() < 0.5:
        put("wordlength", rng.choice([3, 4, 5, 6, 7, 8, 8, 9, 10, 12]), "--wordlength")      # > 8: tagged postings on the device (r03; the host restatement before)
    if rng.random() < 0.2:
        put("minwordmatches", rng.choice([0, 3, 8, 20]), "--minwordmatches")
    if rng.random() < 0.5:
        put("iddef", rng.choice([0, 1, 2, 3, 4]), "--iddef")
    if rng.random() < 0.3:
        put("weak_id", rng.choice([0.4, 0.6, 0.75]), "--weak_id")
    if rng.random() < 0.35:
        o["strand_both"] = 1
        cli += ["--strand", "both"]
    dbm, qm = rng.choice(MASKS), None
    if rng.random() < 0.3:
        qm = rng.choice(MASKS)
    o["soft_mask"] = MASKS.index(dbm)
    if qm is not None:
        o["qmask"] = 1 + MASKS.index(qm)
    cli += ["--dbmask", dbm, "--qmask", qm if qm is not None else dbm]
    for key, flag, vals in (("maxgaps", "--maxgaps", [0, 1, 3, 10]), ("maxsubs", "--maxsubs", [2, 10, 40]), ("maxdiffs", "--maxdiffs", [3, 15, 50]),
                            ("mincols", "--mincols", [50, 120, 200]), ("query_cov", "--query_cov", [0.5, 0.8, 0.95]),
                            ("target_cov", "--target_cov", [0.2, 0.5, 0.9]), ("maxid", "--maxid", [0.9, 0.97, 0.99]), ("mid", "--mid", [80.0, 92.0]),
                            ("minqt", "--minqt", [0.3, 0.6]), ("maxqt", "--maxqt", [0.8, 1.5]), ("minsl", "--minsl", [0.3, 0.6]),
                            ("maxsl", "--maxsl", [0.8, 1.0]), ("idprefix", "--idprefix", [2, 8]), ("idsuffix", "--idsuffix", [2, 8])):
        if rng.random() < 0.12:
            put(key, rng.choice(vals), flag)
    for key, flag in (("leftjust", "--leftjust"), ("rightjust", "--rightjust"), ("selfid", "--selfid")):
        if rng.random() < 0.08:
            o[key] = 1
            cli.append(flag)
    sizes = rng.random() < 0.3
    if sizes:
        cli.append("--sizein")
        for key, flag, vals in (("maxqsize", "--maxqsize", [5, 40]), ("mintsize", "--mintsize", [2, 5]),
                                ("minsizeratio", "--minsizeratio", [0.2, 0.5]), ("maxsizeratio", "--maxsizeratio", [0.5, 2.0, 8.0])):
            if rng.random() < 0.4:
                put(key, rng.choice(vals), flag)
    use_self = rng.random() < 0.15
    if use_self:
        o["self_"] = 1
        cli.append("--self")
    scoring = None
    if rng.random() < 0.4:
        match, mism = rng.randint(1, 5), -rng.randint(1, 8)
        e_i, e_e = rng.randint(1, 4), rng.randint(1, 4)
        o_i, o_e = e_i + rng.randint(0, 24), e_e + rng.randint(0, 10)
        cli += ["--match", str(match), "--mismatch", str(mism), "--gapopen", f"{o_i}I/{o_e}E", "--gapext", f"{e_i}I/{e_e}E"]
        # post-fixup (vsearch.cc:250-259): open -= extension; search16_init order q_l t_l q_i t_i q_r t_r
        scoring = (match, mism, o_e - e_e, o_e - e_e, o_i - e_i, o_i - e_i, o_e - e_e, o_e - e_e, e_e, e_e, e_i, e_i, e_e, e_e)
    return o, scoring, cli, sizes, use_self


def draw_data(rng, sizes, use_self):
    lower = rng.random() < 0.5
    db = M._masked_families(rng, rng.randint(8, 30), rng.randint(3, 8), rng.choice([200, 320, 450]), rng.choice([0.02, 0.06, 0.12]), lower)
    db += [common.rnd_seq(rng, rng.randint(80, 400)) for _ in range(rng.randint(0, 20))]
    if rng.random() < 0.3:
        db += [common.mutate(rng, db[rng.randrange(len(db))], 0.03, "ACGTN") for _ in range(3)]
    qs = M._queries(rng, db, rng.randint(30, 90), rng.choice([90, 150, 220]), rng.choice([0.02, 0.05]), lower)
    qs += [common.mutate(rng, db[rng.randrange(len(db))].upper(), 0.04) for _ in range(10)]
    qs += [db[rng.randrange(len(db))] for _ in range(4)]                      # exact copies (identity 100, --selfid material)
    if rng.random() < 0.06:     # pairs beyond the 16-bit aligner's size guard (Q x D > 25e6): the sentinel and the linear-memory fallback
        anc = common.rnd_seq(rng, rng.randint(5200, 6400))
        db += [common.mutate(rng, anc, 0.03), common.mutate(rng, anc, 0.06)]
        qs += [common.mutate(rng, anc, 0.02), common.mutate(rng, anc[300:5600], 0.04)]
    if rng.random() < 0.5:
        comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "a": "t", "c": "g", "g": "c", "t": "a", "n": "n"}
        for k in range(0, len(qs), 3):
            qs[k] = "".join(comp.get(c, "N") for c in reversed(qs[k]))
    tsize = [rng.choice([1, 1, 2, 3, 8, 30]) for _ in db] if sizes else None
    qsize = [rng.choice([1, 2, 5, 12, 50]) for _ in qs] if sizes else None
    tn = [f"t{i}" + (f";size={tsize[i]}" if sizes else "") for i in range(len(db))]
    qn = [f"q{i}" + (f";size={qsize[i]}" if sizes else "") for i in range(len(qs))]
    if use_self:                                                                 # some queries carry a target's label (and, with
        for k in range(0, len(qs), 5):                                           # --sizein, the abundance written in it)
            t = rng.randrange(len(tn))
            qn[k] = tn[t]
            if sizes:
                qsize[k] = tsize[t]
    return db, qs, tn, qn, tsize, qsize


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--max-rounds", type=int, default=0, help="stop after this many rounds (0: run for --seconds): a deterministic set of rounds for a given seed")
    ap.add_argument("--only-round", type=int, default=-1, help="replay the draws, run just this round and print its whole diff")
    ap.add_argument("--wordlengths", default="", help="LO..HI: draw every round's --wordlength from this range (e.g. 9..15: the tagged device index)")
    a = ap.parse_args()
    if a.wordlengths:
        global WORDLENGTHS
        lo, hi = a.wordlengths.split("..")
        WORDLENGTHS = (int(lo), int(hi))
    if not refcli.available():
        raise SystemExit("oracle/_ref/vsearch_ref missing: make -C oracle ref_full")
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    rounds = lines = bad_rounds = 0
    failing = []
    with tempfile.TemporaryDirectory(prefix="vsxsoak_") as tmp:
        dbf, qf, uf = (os.path.join(tmp, x) for x in ("db.fa", "q.fa", "u.tsv"))
        while time.time() < t_end and (a.max_rounds <= 0 or rounds < a.max_rounds):
            o, scoring, cli, sizes, use_self = draw_options(rng)
            db, qs, tn, qn, tsize, qsize = draw_data(rng, sizes, use_self)
            if a.only_round >= 0 and rounds != a.only_round:
                rounds += 1
                if rounds > a.only_round:
                    break
                continue
            refcli.write_fasta(dbf, tn, db)
            refcli.write_fasta(qf, qn, qs)
            p = subprocess.run([refcli.REF_BIN, "--usearch_global", qf, "--db", dbf, "--threads", "1", "--userout", uf, "--userfields", "+".join(FIELDS),
                                "--quiet"] + cli, capture_output=True, text=True)
            if p.returncode != 0:
                failing.append({"cli": cli, "error": p.stderr[-400:]})
                bad_rounds += 1
                rounds += 1
                continue
            exp = open(uf).read().splitlines()
            with (Aligner(scoring=scoring) if scoring else Aligner()) as al:
                ss = SearchSession(al, db, sizes=tsize, labels=tn if (sizes or use_self) else None, **o)
                hits = ss.search_batch(qs, sizes=qsize, labels=qn if (sizes or use_self) else None)
                got = ss.userout(qs, qnames=qn, tnames=tn, fields=FIELDS, hits=hits)
            rounds += 1
            lines += len(exp)
            if a.only_round >= 0:
                import difflib
                print("cli:", " ".join(cli))
                for ln in difflib.unified_diff(exp, got, "reference", "vsx", lineterm="", n=0):
                    print(ln[:260])
                keep = os.path.join(ROOT, "gpurun_out", f"soak_round{a.only_round}")
                os.makedirs(keep, exist_ok=True)
                refcli.write_fasta(os.path.join(keep, "db.fa"), tn, db)
                refcli.write_fasta(os.path.join(keep, "q.fa"), qn, qs)
                json.dump({"opts": o, "scoring": scoring, "cli": cli}, open(os.path.join(keep, "round.json"), "w"))
            if got != exp:
                bad_rounds += 1
                if len(failing) < 12:
                    first = next((i for i, (x, y) in enumerate(zip(got, exp)) if x != y), min(len(got), len(exp)))
                    failing.append({"cli": cli, "opts": {k: v for k, v in o.items()}, "scoring": scoring, "lines": [len(got), len(exp)], "first_diff": first,
                                    "got": got[first] if first < len(got) else None, "exp": exp[first] if first < len(exp) else None,
                                    "data_seed_hint": [a.seed, rounds]})
    out = {"rounds": rounds, "userout_lines": lines, "failing_rounds": bad_rounds, "failures": failing, "seed": a.seed, "seconds": a.seconds,
           "what": "vsx_search_batch (SearchSession.userout, 15 fields) vs vsearch_ref --usearch_global --userout with the same randomly drawn options"}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f)
    sys.exit(1 if bad_rounds else 0)


if __name__ == "__main__":
    main()
