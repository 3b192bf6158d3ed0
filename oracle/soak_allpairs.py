#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: randomized soak of vsx_allpairs_rows (pair enumeration with the unaligned filters, pipelined plans, the
accept filter and the ranking / compaction of kept hits ON THE DEVICE) against the REFERENCE CLI (vsearch_ref --allpairs_global
--userout), byte for byte.  Options per round: --acceptall or --id, the aligned / unaligned filters, identity definition, abundance
filters with --sizein, --self, a scoring set in the CLI's syntax; blocks of random size as a caller would walk the database.

    python oracle/soak_allpairs.py --seconds 120 --seed 1 --out gpurun_out/soak_allpairs.json
"""
import argparse
import json
import os
os.environ.setdefault("VSX_RANK_STRICT", "1")      # any device/host difference in identity or filter is an error here
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

FIELDS = ["query", "target", "id", "alnlen", "mism", "opens", "exts", "raw", "caln", "id0", "id1", "id2", "id3", "id4"]


def draw(rng):
    o, cli = {}, []

    def put(key, val, flag):
        o[key] = val
        cli.extend([flag, repr(val) if isinstance(val, float) else str(val)])

    acceptall = rng.random() < 0.25
    if acceptall:
        cli.append("--acceptall")
        o["id"] = 0.0
    else:
        put("id", rng.choice([0.5, 0.7, 0.8, 0.9, 0.97]), "--id")
    if rng.random() < 0.5:
        put("iddef", rng.choice([0, 1, 2, 3, 4]), "--iddef")
    for key, flag, vals in (("maxgaps", "--maxgaps", [0, 2, 8]), ("maxsubs", "--maxsubs", [3, 20]), ("maxdiffs", "--maxdiffs", [5, 30]),
                            ("mincols", "--mincols", [60, 150]), ("query_cov", "--query_cov", [0.6, 0.9]), ("target_cov", "--target_cov", [0.6, 0.9]),
                            ("maxid", "--maxid", [0.95, 0.99]), ("mid", "--mid", [85.0]), ("minqt", "--minqt", [0.5, 0.8]), ("maxqt", "--maxqt", [1.2]),
                            ("minsl", "--minsl", [0.5, 0.8]), ("maxsl", "--maxsl", [0.9]), ("idprefix", "--idprefix", [3]), ("idsuffix", "--idsuffix", [3])):
        if rng.random() < 0.1:
            put(key, rng.choice(vals), flag)
    for key, flag in (("leftjust", "--leftjust"), ("rightjust", "--rightjust"), ("selfid", "--selfid")):
        if rng.random() < 0.06:
            o[key] = 1
            cli.append(flag)
    sizes = rng.random() < 0.3
    if sizes:
        cli.append("--sizein")
        for key, flag, vals in (("maxqsize", "--maxqsize", [5, 40]), ("mintsize", "--mintsize", [2, 5]),
                                ("minsizeratio", "--minsizeratio", [0.2, 0.5]), ("maxsizeratio", "--maxsizeratio", [0.5, 2.0])):
            if rng.random() < 0.35:
                put(key, rng.choice(vals), flag)
    if rng.random() < 0.12:        # --self: pairs whose labels coincide are skipped (the data below then repeats some labels)
        o["self_"] = 1
        cli.append("--self")
    scoring = None
    if rng.random() < 0.4:
        match, mism = rng.randint(1, 5), -rng.randint(1, 8)
        e_i, e_e = rng.randint(1, 4), rng.randint(1, 4)
        o_i, o_e = e_i + rng.randint(0, 24), e_e + rng.randint(0, 10)
        cli += ["--match", str(match), "--mismatch", str(mism), "--gapopen", f"{o_i}I/{o_e}E", "--gapext", f"{e_i}I/{e_e}E"]
        scoring = (match, mism, o_e - e_e, o_e - e_e, o_i - e_i, o_i - e_i, o_e - e_e, o_e - e_e, e_e, e_e, e_i, e_i, e_e, e_e)
    return o, scoring, cli, acceptall, sizes


def data(rng, sizes, dup_labels=False):
    seqs = M._masked_families(rng, rng.randint(3, 10), rng.randint(2, 8), rng.choice([100, 220, 330]), rng.choice([0.02, 0.06, 0.15]), rng.random() < 0.3)
    seqs += [common.rnd_seq(rng, rng.randint(40, 300)) for _ in range(rng.randint(0, 8))]
    seqs += [seqs[rng.randrange(len(seqs))] for _ in range(rng.randint(0, 3))]
    if rng.random() < 0.2:
        seqs += [common.mutate(rng, seqs[0].upper(), 0.05, "ACGTNRY")]
    rng.shuffle(seqs)
    sz = [rng.choice([1, 1, 2, 3, 8, 30]) for _ in seqs] if sizes else None
    names = [f"t{i}" + (f";size={sz[i]}" if sizes else "") for i in range(len(seqs))]
    if dup_labels:
        for k in range(1, len(names), 4):
            names[k] = names[k - 1]
            if sizes:
                sz[k] = sz[k - 1]
    return seqs, names, sz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--max-rounds", type=int, default=0, help="stop after this many rounds (0: run for --seconds): a deterministic set of rounds for a given seed")
    a = ap.parse_args()
    if not refcli.available():
        raise SystemExit("oracle/_ref/vsearch_ref missing: make -C oracle ref_full")
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    rounds = lines = bad = 0
    failing = []
    with tempfile.TemporaryDirectory(prefix="vsxsoaka_") as tmp:
        fa, uo = os.path.join(tmp, "a.fa"), os.path.join(tmp, "u.tsv")
        while time.time() < t_end and (a.max_rounds <= 0 or rounds < a.max_rounds):
            o, scoring, cli, acceptall, sizes = draw(rng)
            seqs, names, sz = data(rng, sizes, bool(o.get("self_")))
            block = rng.choice([1, 5, 16, 1000])
            refcli.write_fasta(fa, names, seqs)
            p = subprocess.run([refcli.REF_BIN, "--allpairs_global", fa, "--qmask", "none", "--threads", "1", "--userout", uo, "--userfields", "+".join(FIELDS),
                                "--quiet"] + cli, capture_output=True, text=True)
            rounds += 1
            if p.returncode != 0:
                bad += 1
                failing.append({"cli": cli, "error": p.stderr[-300:]})
                continue
            exp = open(uo).read().splitlines()
            with (Aligner(scoring=scoring) if scoring else Aligner()) as al:
                ss = SearchSession(al, seqs, sizes=sz, labels=names if (sizes or o.get("self_")) else None, **o)
                hits = []
                for first in range(0, len(seqs), block):
                    hits += ss.allpairs(first, min(block, len(seqs) - first), acceptall=acceptall)
                got = ss.userout(seqs, qnames=names, tnames=names, fields=FIELDS, hits=hits)
            lines += len(exp)
            if got != exp:
                bad += 1
                if len(failing) < 10:
                    first = next((i for i, (x, y) in enumerate(zip(got, exp)) if x != y), min(len(got), len(exp)))
                    failing.append({"cli": cli, "scoring": scoring, "n": len(seqs), "block": block, "lines": [len(got), len(exp)], "first_diff": first,
                                    "got": got[first] if first < len(got) else None, "exp": exp[first] if first < len(exp) else None, "round": rounds - 1})
    out = {"rounds": rounds, "userout_lines": lines, "failing_rounds": bad, "failures": failing, "seed": a.seed, "seconds": a.seconds,
           "what": "vsx_allpairs_block (device filter + device ranking) vs vsearch_ref --allpairs_global --userout with the same randomly drawn options"}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
