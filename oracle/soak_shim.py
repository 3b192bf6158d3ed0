#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: randomized soak of the drop-in boundary.  oracle/_ref/vsearch_vsx is the reference CLI with exactly one
translation unit swapped (core/align_simd.cpp -> shim/vsx_search16_shim.cpp + libvsx.so); every round runs one of the commands
that reach the aligner (--usearch_global, --cluster_fast, --cluster_size, --allpairs_global, --uchime_ref) with randomly drawn
options, scoring and thread count through BOTH binaries and compares every output file byte for byte (--alnout, --userout,
--uc, --samout, --msaout, --consout, --profile, --uchimeout, --uchimealns, ...).

    python oracle/soak_shim.py --seconds 120 --seed 1 --out gpurun_out/soak_shim.json
"""
import argparse
import json
import os
import random
import re
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

VSX_BIN = os.path.join(HERE, "_ref", "vsearch_vsx")
FIELDS = "query+target+id+alnlen+mism+opens+exts+raw+caln+qlo+qhi+tlo+thi+id0+id1+id2+id3+id4+qrow+trow"


def scoring_args(rng):
    if rng.random() < 0.5:
        return []
    e_i, e_e = rng.randint(1, 4), rng.randint(1, 4)
    return ["--match", str(rng.randint(1, 5)), "--mismatch", str(-rng.randint(1, 8)), "--gapopen", f"{e_i + rng.randint(0, 24)}I/{e_e + rng.randint(0, 10)}E",
            "--gapext", f"{e_i}I/{e_e}E"]


def draw(rng, tmp):
    """-> (argv tail as a function of the output directory, [output files])"""
    cmd = rng.choice(["usearch_global", "usearch_global", "cluster_fast", "cluster_size", "allpairs_global", "uchime_ref"])
    mask = rng.choice(["none", "none", "soft", "dust"])
    sc = scoring_args(rng)
    lower = rng.random() < 0.3
    names = lambda p, seqs: [f"{p}{i};size={1 + (i * 7) % 5}" for i in range(len(seqs))]
    if cmd == "usearch_global":
        db = M._masked_families(rng, rng.randint(5, 14), rng.randint(3, 7), rng.choice([200, 380]), rng.choice([0.03, 0.08]), lower)
        db += [common.rnd_seq(rng, rng.randint(100, 400)) for _ in range(rng.randint(0, 10))]
        qs = M._queries(rng, db, rng.randint(20, 60), rng.choice([100, 170]), 0.04, lower) + [common.mutate(rng, db[0].upper(), 0.05, "ACGTNRY")]
        refcli.write_fasta(tmp + "/db.fa", names("t", db), db)
        refcli.write_fasta(tmp + "/q.fa", names("q", qs), qs)
        extra = ["--id", str(rng.choice([0.6, 0.8, 0.9, 0.97])), "--maxaccepts", str(rng.choice([0, 1, 3])), "--maxrejects", str(rng.choice([0, 8, 32]))]
        if rng.random() < 0.4:
            extra += ["--strand", "both"]
        if rng.random() < 0.3:
            extra += ["--iddef", str(rng.randint(0, 4))]
        files = ["u.tsv", "aln.txt", "hits.uc", "hits.sam", "b6.txt"]
        return cmd, lambda out: (["--usearch_global", tmp + "/q.fa", "--db", tmp + "/db.fa", "--qmask", mask, "--dbmask", mask, "--userout", out + "/u.tsv",
                                  "--userfields", FIELDS, "--alnout", out + "/aln.txt", "--uc", out + "/hits.uc", "--samout", out + "/hits.sam",
                                  "--blast6out", out + "/b6.txt"] + extra + sc), files
    if cmd in ("cluster_fast", "cluster_size"):
        seqs = M._masked_families(rng, rng.randint(4, 12), rng.randint(2, 9), rng.choice([150, 300]), rng.choice([0.02, 0.05]), lower)
        seqs += [common.rnd_seq(rng, rng.randint(100, 320)) for _ in range(rng.randint(0, 8))]
        rng.shuffle(seqs)
        refcli.write_fasta(tmp + "/in.fa", names("s", seqs), seqs)
        extra = ["--id", str(rng.choice([0.85, 0.9, 0.97])), "--sizein", "--sizeout"]
        if rng.random() < 0.4:
            extra += ["--maxaccepts", str(rng.choice([2, 3]))]
        if cmd == "cluster_size" and rng.random() < 0.5:
            extra += ["--sizeorder"]
        files = ["c.uc", "cent.fa", "msa.fa", "cons.fa", "prof.txt"]
        return cmd, lambda out: (["--" + cmd, tmp + "/in.fa", "--qmask", mask, "--uc", out + "/c.uc", "--centroids", out + "/cent.fa", "--msaout", out + "/msa.fa",
                                  "--consout", out + "/cons.fa", "--profile", out + "/prof.txt"] + extra + sc), files
    if cmd == "allpairs_global":
        seqs = M._masked_families(rng, rng.randint(2, 6), rng.randint(2, 7), rng.choice([120, 260]), rng.choice([0.04, 0.1]), lower)
        refcli.write_fasta(tmp + "/in.fa", names("s", seqs), seqs)
        extra = ["--acceptall"] if rng.random() < 0.3 else ["--id", str(rng.choice([0.6, 0.75, 0.9]))]
        files = ["u.tsv", "aln.txt"]
        return cmd, lambda out: (["--allpairs_global", tmp + "/in.fa", "--qmask", mask, "--userout", out + "/u.tsv", "--userfields", FIELDS,
                                  "--alnout", out + "/aln.txt"] + extra + sc), files
    parents = [common.rnd_seq(rng, rng.choice([300, 400])) for _ in range(rng.randint(4, 8))]
    qs = []
    for _ in range(rng.randint(6, 14)):
        a, b = rng.sample(range(len(parents)), 2)
        cut = rng.randint(100, 200)
        qs.append(common.mutate(rng, parents[a][:cut] + parents[b][cut:], 0.01))
    qs += [common.mutate(rng, p, 0.02) for p in parents[:3]]
    refcli.write_fasta(tmp + "/ref.fa", names("p", parents), parents)
    refcli.write_fasta(tmp + "/q.fa", names("q", qs), qs)
    files = ["uchime.tsv", "alns.txt", "chim.fa", "non.fa"]
    return cmd, lambda out: (["--uchime_ref", tmp + "/q.fa", "--db", tmp + "/ref.fa", "--qmask", mask, "--dbmask", mask, "--uchimeout", out + "/uchime.tsv",
                              "--uchimealns", out + "/alns.txt", "--chimeras", out + "/chim.fa", "--nonchimeras", out + "/non.fa"]), files


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--max-rounds", type=int, default=0, help="stop after this many rounds (0: run for --seconds): a deterministic set of rounds for a given seed")
    a = ap.parse_args()
    if not (refcli.available() and os.path.exists(VSX_BIN)):
        raise SystemExit("oracle/_ref/vsearch_ref / vsearch_vsx missing: make -C oracle ref_full ref_shim")
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    rounds = bad = files_ok = 0
    by_cmd, failing = {}, []
    with tempfile.TemporaryDirectory(prefix="vsxsoaks_") as tmp:
        while time.time() < t_end and (a.max_rounds <= 0 or rounds < a.max_rounds):
            cmd, argv_of, files = draw(rng, tmp)
            threads = rng.choice(["1", "1", "3"])
            res = {}
            for tag, exe in (("ref", refcli.REF_BIN), ("vsx", VSX_BIN)):
                out = os.path.join(tmp, tag)
                os.makedirs(out, exist_ok=True)
                for f in os.listdir(out):
                    os.remove(os.path.join(out, f))
                # several reference threads write hits in completion order: only the single-threaded reference is canonical
                p = subprocess.run([exe] + argv_of(out) + ["--threads", "1" if tag == "ref" else threads, "--quiet"], capture_output=True, text=True, timeout=600)
                if p.returncode != 0:
                    res[tag] = {"error": p.stderr[-300:]}
                    continue
                res[tag] = {}
                for f in files:
                    data = open(os.path.join(out, f), "rb").read() if os.path.exists(os.path.join(out, f)) else b"<missing>"
                    # --alnout / --uchimealns echo the command line: neutralise the binary, the output directory, the thread count
                    res[tag][f] = re.sub(rb"--threads \d+", b"--threads N", data.replace(exe.encode(), b"VSEARCH").replace(out.encode(), b"OUT"))
            rounds += 1
            by_cmd[cmd] = by_cmd.get(cmd, 0) + 1
            diff = []
            if "error" in res["ref"] or "error" in res["vsx"]:
                diff = ["error", res["ref"].get("error"), res["vsx"].get("error")]
            else:
                for f in files:
                    x, y = res["ref"][f], res["vsx"][f]
                    if threads != "1" and cmd != "cluster_fast" and cmd != "cluster_size":
                        x, y = b"\n".join(sorted(x.split(b"\n"))), b"\n".join(sorted(y.split(b"\n")))
                    if x != y:
                        diff.append(f)
                    else:
                        files_ok += 1
            if diff:
                bad += 1
                if len(failing) < 10:
                    failing.append({"cmd": cmd, "argv": argv_of("OUT"), "threads": threads, "differ": [str(d)[:300] for d in diff], "round": rounds - 1})
    out = {"rounds": rounds, "by_command": by_cmd, "files_identical": files_ok, "failing_rounds": bad, "failures": failing, "seed": a.seed, "seconds": a.seconds,
           "what": "vsearch_vsx (reference CLI with core/align_simd.cpp swapped for the shim + libvsx) vs vsearch_ref: every output file"}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
