#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: randomized soak of vsx_cluster_fast (rounds on the GPU + the intra-round fix-up) against the
REFERENCE CLI (oracle/_ref/vsearch_ref --cluster_fast / --cluster_size), --uc files byte for byte.

Every round draws the command (length- or abundance-sorted, --sizeorder), --id, maxaccepts / maxrejects, word length, identity
definition, masking, a scoring set in the CLI's syntax, the ROUND SIZE of the GPU stages (the result must not depend on it) and a
family-structured data set (amplicon-like, duplicates, low-complexity stretches).

    python oracle/soak_cluster.py --seconds 120 --seed 1 --out gpurun_out/soak_cluster.json
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

MASKS = ["none", "soft", "dust"]


def draw(rng):
    o, cli = {}, []

    def put(key, val, flag):
        o[key] = val
        cli.extend([flag, repr(val) if isinstance(val, float) else str(val)])

    by_size = rng.random() < 0.35
    unoise = by_size and rng.random() < 0.3        # --cluster_unoise: abundance order, the skew rule instead of --id, weak_id forced to 0.90
    if unoise:
        alpha = rng.choice([1.0, 2.0, 3.0])
        o.update(id=0.0, maxaccepts=1, maxrejects=32, cluster_unoise=1, unoise_alpha=alpha)
        cli += ["--minsize", "1", "--sizein", "--unoise_alpha", repr(alpha)]
        mask = rng.choice(MASKS)
        o["soft_mask"] = MASKS.index(mask)
        cli += ["--qmask", mask]
        return o, None, cli, "unoise", rng.choice([3, 16, 50, 100000])
    put("id", rng.choice([0.8, 0.9, 0.95, 0.97, 0.99]), "--id")
    put("maxaccepts", rng.choice([1, 1, 2, 3]), "--maxaccepts")
    put("maxrejects", rng.choice([2, 8, 8, 16, 32]), "--maxrejects")
    if rng.random() < 0.4:
        put("wordlength", rng.choice([5, 6, 7, 8]), "--wordlength")
    if rng.random() < 0.4:
        put("iddef", rng.choice([0, 1, 2, 3, 4]), "--iddef")
    if rng.random() < 0.2:
        put("maxgaps", rng.choice([1, 4]), "--maxgaps")
    if rng.random() < 0.15:
        put("query_cov", rng.choice([0.8, 0.95]), "--query_cov")
    mask = rng.choice(MASKS)
    o["soft_mask"] = MASKS.index(mask)
    cli += ["--qmask", mask]
    if by_size:
        cli.append("--sizein")
        if rng.random() < 0.5:
            o["sizeorder"] = 1
            cli.append("--sizeorder")
    scoring = None
    if rng.random() < 0.3:
        match, mism = rng.randint(1, 4), -rng.randint(1, 7)
        e_i, e_e = rng.randint(1, 3), rng.randint(1, 3)
        o_i, o_e = e_i + rng.randint(0, 20), e_e + rng.randint(0, 6)
        cli += ["--match", str(match), "--mismatch", str(mism), "--gapopen", f"{o_i}I/{o_e}E", "--gapext", f"{e_i}I/{e_e}E"]
        scoring = (match, mism, o_e - e_e, o_e - e_e, o_i - e_i, o_i - e_i, o_e - e_e, o_e - e_e, e_e, e_e, e_i, e_i, e_e, e_e)
    round_size = rng.choice([3, 7, 16, 50, 200, 100000])
    return o, scoring, cli, by_size, round_size


def data(rng, by_size):
    lower = rng.random() < 0.3
    seqs = M._masked_families(rng, rng.randint(6, 25), rng.randint(2, 12), rng.choice([120, 250, 320]), rng.choice([0.01, 0.03, 0.08]), lower)
    seqs += [common.rnd_seq(rng, rng.randint(60, 340)) for _ in range(rng.randint(0, 15))]
    for _ in range(rng.randint(0, 6)):                                         # duplicates and near-duplicates
        s = seqs[rng.randrange(len(seqs))]
        seqs.append(s if rng.random() < 0.5 else s[:-rng.randint(1, 4)])
    rng.shuffle(seqs)
    sz = [rng.choice([1, 1, 1, 2, 3, 5, 9, 30, 200]) for _ in seqs] if by_size else None
    names = [f"s{i:04d}" + (f";size={sz[i]}" if by_size else "") for i in range(len(seqs))]
    if by_size:     # Database::sortbyabundance (core/db.cpp:471-486): abundance descending, then label, then input order
        order = sorted(range(len(seqs)), key=lambda i: (-sz[i], names[i], i))
    else:           # Database::sortbylength (core/db.cpp:433-450): length descending, abundance, label
        order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), names[i]))
    return seqs, names, sz, order


def _wrap(seq, width=80):
    return [seq[i:i + width] for i in range(0, len(seq), width)] or [""]


def msa_files(ss, al, sseqs, snames, round_size):
    """--msaout / --consout / --profile lines from vsx_cluster_fast + vsx_msa_device_batch (format: core/msa.cpp:390-560)"""
    from vsearch_amd import msa_batch
    cno, per, ncl = ss.cluster_fast(round=round_size)
    members = [[] for _ in range(ncl)]
    for s, c in enumerate(cno):
        members[c].append(s)
    results = msa_batch([[sseqs[s] for s in m] for m in members], [[None] + [per[s]["cigar"] for s in m[1:]] for m in members], None, al)
    msa_lines, cons_lines, prof_lines = [], [], []
    for c in range(ncl):
        m, res = members[c], results[c]
        msa_lines.append("")
        for k, s in enumerate(m):
            msa_lines.append(">" + ("*" if k == 0 else "") + snames[s])
            msa_lines += _wrap(res["rows"][k])
        msa_lines.append(">consensus")
        msa_lines += _wrap(res["rows"][-1])
        cons_lines.append(f">centroid={snames[m[0]]};seqs={len(m)}")
        cons_lines += _wrap(res["consensus"])
        prof_lines.append(f">centroid={snames[m[0]]};seqs={len(m)}")
        for i, (ch, pr) in enumerate(zip(res["rows"][-1], res["profile"])):
            prof_lines.append("\t".join([str(i), ch] + [str(pr[k]) for k in (0, 1, 2, 3, 5, 4)]))
        prof_lines.append("")
    return msa_lines, cons_lines, prof_lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--max-rounds", type=int, default=0, help="stop after this many rounds (0: run for --seconds): a deterministic set of rounds for a given seed")
    ap.add_argument("--force-round-size", type=int, default=0, help="stress: every round of the soak uses this GPU round size (3: hundreds of tiny rounds per data set)")
    ap.add_argument("--only-by-size", action="store_true", help="stress: only the abundance-sorted commands (--cluster_size / --cluster_unoise)")
    ap.add_argument("--only-round", type=int, default=-1, help="replay: draw every round of the seed but run only this one (and print its first difference)")
    a = ap.parse_args()
    if not refcli.available():
        raise SystemExit("oracle/_ref/vsearch_ref missing: make -C oracle ref_full")
    from vsearch_amd import Aligner, SearchSession
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    rounds = lines = bad = msa_rounds = 0
    failing = []
    with tempfile.TemporaryDirectory(prefix="vsxsoakc_") as tmp:
        fa, uc = os.path.join(tmp, "c.fa"), os.path.join(tmp, "c.uc")
        while time.time() < t_end and (a.max_rounds <= 0 or rounds < a.max_rounds):
            o, scoring, cli, by_size, round_size = draw(rng)
            if a.force_round_size:
                round_size = a.force_round_size
            unoise = by_size == "unoise"
            seqs, names, sz, order = data(rng, bool(by_size))
            if a.only_by_size and not by_size:
                continue
            refcli.write_fasta(fa, names, seqs)
            # the CIGAR consumer too (msa.cpp): star MSA, consensus and profile of every cluster, on the device, in half of the
            # cluster_fast rounds without masking (masking changes the case of the printed rows)
            want_msa = (not by_size) and o["soft_mask"] == 0 and rng.random() < 0.5
            if a.only_round >= 0:
                if rounds < a.only_round:
                    rounds += 1
                    continue
                if rounds > a.only_round:
                    break
            msa_args = ["--msaout", tmp + "/m.msa", "--consout", tmp + "/m.cons", "--profile", tmp + "/m.prof"] if want_msa else []
            p = subprocess.run([refcli.REF_BIN, "--cluster_unoise" if unoise else ("--cluster_size" if by_size else "--cluster_fast"), fa, "--threads", "1", "--uc", uc, "--quiet"] + cli + msa_args,
                               capture_output=True, text=True)
            rounds += 1
            if p.returncode != 0:
                bad += 1
                failing.append({"cli": cli, "error": p.stderr[-300:]})
                continue
            exp = open(uc).read().splitlines()
            sseqs, snames = [seqs[i] for i in order], [names[i] for i in order]
            ssz = [sz[i] for i in order] if by_size else None
            with (Aligner(scoring=scoring) if scoring else Aligner()) as al:
                ss = SearchSession(al, sseqs, sizes=ssz, labels=snames if by_size else None, **o)
                got = ss.uc_lines(snames, round=round_size, sizes=ssz, command="cluster_unoise" if unoise else ("cluster_size" if by_size else "cluster_fast"))
                if want_msa:
                    got_msa = msa_files(ss, al, sseqs, snames, round_size)
                    exp_msa = tuple(open(tmp + x).read().splitlines() for x in ("/m.msa", "/m.cons", "/m.prof"))
                    msa_rounds += 1
                    if got_msa != exp_msa:
                        got = got + ["<msa / consensus / profile differ>"]
            lines += len(exp)
            if got != exp:
                bad += 1
                if len(failing) < 10:
                    first = next((i for i, (x, y) in enumerate(zip(got, exp)) if x != y), min(len(got), len(exp)))
                    failing.append({"cli": cli, "by_size": by_size, "round_size": round_size, "scoring": scoring, "n": len(seqs), "lines": [len(got), len(exp)],
                                    "first_diff": first, "got": got[first] if first < len(got) else None, "exp": exp[first] if first < len(exp) else None,
                                    "round": rounds - 1})
    out = {"rounds": rounds, "msa_rounds": msa_rounds, "uc_lines": lines, "failing_rounds": bad, "failures": failing, "seed": a.seed, "seconds": a.seconds,
           "what": "vsx_cluster_fast (SearchSession.uc_lines) vs vsearch_ref --cluster_fast / --cluster_size --uc with the same randomly drawn options"}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
