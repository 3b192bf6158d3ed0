#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libvsref.so, built from
/root/reference/src by oracle/Makefile) -- run in the build container only (the GPU box has
no /root/reference).  The fixtures pin both oracle/nw_oracle.c and the HIP path.

  python oracle/gen_golden.py            # writes tests/golden/search16_golden.json,
                                         #        tests/golden/ref_api_examples.json, lma_golden.json, dust_golden.json
  python oracle/gen_golden.py --fuzz N   # additionally: N random pairs oracle-vs-reference (no file)
"""
import argparse
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# scoring sets: (name, P[14] in search16_init order, n_mismatch)
SCORINGS = [
    ("default", pyoracle.DEFAULT_P, False),
    ("distinct12", (3, -5, 3, 7, 11, 13, 2, 5, 1, 2, 3, 4, 2, 1), False),
    ("nmismatch", pyoracle.DEFAULT_P, True),
    ("zero_terminal", (2, -4, 0, 0, 18, 18, 0, 0, 0, 0, 2, 2, 0, 0), False),
    ("uniform10_1", (1, -2, 10, 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1), False),
    ("big_left_ext", (2, -4, 1, 1, 18, 18, 1, 1, 30, 40, 2, 2, 1, 1), False),
    ("pos_overflow", (300, -400, 1, 1, 18, 18, 1, 1, 1, 1, 2, 2, 1, 1), False),
    ("neg_overflow", (2, -3000, 200, 300, 2000, 3000, 200, 300, 200, 300, 400, 500, 200, 300), False),
    ("all6553", (2, -4) + (6553,) * 12, False),
    ("forced_fallback", (2, -4, 1, 1, 6554, 18, 1, 1, 1, 1, 2, 2, 1, 1), False),
    ("star_penalty", (2, -4, 1, 1, 2147483647, 18, 1, 1, 1, 1, 2, 2, 1, 1), False),
]

IUPAC = "ACGTURYSWKMBDHVN"


def rnd_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, rate, alphabet="ACGT"):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate * 0.8:
            out.append(rng.choice(alphabet))
        elif r < rate * 0.9:
            continue
        elif r < rate:
            out.append(ch)
            out.append(rng.choice(alphabet))
        else:
            out.append(ch)
    return "".join(out)


def make_pair(rng, kind):
    if kind == "related":
        L = rng.randint(1, 300)
        a = rnd_seq(rng, L)
        b = mutate(rng, a, rng.choice([0.02, 0.08, 0.2]))
        if rng.random() < 0.5:   # query is a window of a longer target (semi-global shape)
            b = rnd_seq(rng, rng.randint(0, 80)) + b + rnd_seq(rng, rng.randint(0, 80))
        return a, b
    if kind == "unrelated":
        return rnd_seq(rng, rng.randint(1, 120)), rnd_seq(rng, rng.randint(1, 160))
    if kind == "iupac":
        L = rng.randint(1, 150)
        a = rnd_seq(rng, L, IUPAC + "acgtn" + "X-")
        b = mutate(rng, a, 0.1, IUPAC + "acgtn")
        return a, b
    if kind == "tiny":
        return rnd_seq(rng, rng.randint(0, 3)), rnd_seq(rng, rng.randint(0, 6))
    if kind == "gappy":
        L = rng.randint(20, 200)
        a = rnd_seq(rng, L)
        cut = rng.randint(1, L - 1)
        w = rng.randint(1, 40)
        b = a[:cut] + (rnd_seq(rng, w) if rng.random() < 0.5 else "") + a[min(L, cut + (0 if rng.random() < 0.5 else w)):]
        return (a, b) if rng.random() < 0.5 else (b, a)
    raise ValueError(kind)


KINDS = ["related", "related", "unrelated", "iupac", "tiny", "gappy"]


def fuzz(n, seed):
    rng = random.Random(seed)
    orc = pyoracle.Oracle()
    bad = 0
    per = max(1, n // len(SCORINGS))
    for name, P, nmm in SCORINGS:
        ref = pyoracle.Reference(P, nmm)
        sent = 0
        for _ in range(per):
            q, t = make_pair(rng, rng.choice(KINDS))
            r = ref.align(q, t)
            o = orc.align(q, t, P, nmm)
            sent += r[0] == 32767
            if tuple(r) != tuple(o):
                bad += 1
                if bad < 10:
                    print("MISMATCH", name, repr(q), repr(t), r, o)
        ref.close()
        print(f"  {name}: {per} pairs, {sent} sentinels")
    print(f"fuzz: {per * len(SCORINGS)} pairs, {bad} mismatches")
    return bad


def gen_search16(path, seed=20260924, per_scoring=60):
    rng = random.Random(seed)
    cases = []
    for name, P, nmm in SCORINGS:
        ref = pyoracle.Reference(P, nmm)
        for _ in range(per_scoring):
            q, t = make_pair(rng, rng.choice(KINDS))
            r = ref.align(q, t)
            cases.append({"scoring": name, "q": q, "t": t, "exp": list(r)})
        # the reference's own batch shape: one query against several targets in ONE search16 call
        q = rnd_seq(rng, 120)
        ts = [mutate(rng, q, 0.1) for _ in range(11)] + ["", "A", rnd_seq(rng, 333)]
        for t, r in zip(ts, ref.search16(q, ts)):
            cases.append({"scoring": name, "q": q, "t": t, "exp": list(r), "batched": True})
        ref.close()
    # bench-shaped pairs under defaults (250 x ~1000 family-structured)
    ref = pyoracle.Reference()
    for _ in range(12):
        anc = rnd_seq(rng, 1000)
        mem = mutate(rng, anc, 0.08)
        off = rng.randint(0, len(mem) - 250)
        q = mutate(rng, mem[off:off + 250], 0.03)
        t = mutate(rng, anc, 0.08)
        cases.append({"scoring": "default", "q": q, "t": t, "exp": list(ref.align(q, t))})
    ref.close()
    doc = {
        "generator": "oracle/gen_golden.py (reference search16 via oracle/_ref/libvsref.so, vsearch 2.31.0)",
        "scorings": {n: {"P": list(P), "n_mismatch": nmm} for n, P, nmm in SCORINGS},
        "fields": ["score", "aligned", "matches", "mismatches", "gaps", "cigar"],
        "cases": cases,
    }
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path)} bytes")


def read_fasta(path):
    out, name, buf = [], None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(buf)))
                name, buf = line[1:].split()[0], []
            elif line:
                buf.append(line)
    if name is not None:
        out.append((name, "".join(buf)))
    return out


def gen_api_examples(path):
    """The reference's in-tree golden vectors that pin this path (SURVEY.md 8c level 3):
    api_examples/data/expected_search.tsv (query,target,%id) and expected_cluster.uc H rows."""
    d = os.path.join(pyoracle.REFERENCE_ROOT, "api_examples", "data")
    refs = dict(read_fasta(os.path.join(d, "chimera_ref.fasta")))
    qs = dict(read_fasta(os.path.join(d, "chimera_queries.fasta")))
    search = []
    with open(os.path.join(d, "expected_search.tsv")) as f:
        for line in f:
            qn, tn, pid = line.split()
            search.append({"query": qn, "target": tn, "id": pid})
    cluster = []
    with open(os.path.join(d, "expected_cluster.uc")) as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            if c[0] == "H":
                cluster.append({"query": c[8], "target": c[9], "id": c[3], "cigar": c[7], "len": int(c[2])})
    doc = {
        "source": "reference api_examples/data: chimera_ref.fasta, chimera_queries.fasta, "
                  "expected_search.tsv (--id 0.5 --maxaccepts 3 --maxrejects 16), expected_cluster.uc "
                  "(--cluster_fast --id 0.70)",
        "refs": refs, "queries": qs, "expected_search": search, "expected_cluster_hits": cluster,
    }
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(f"wrote {path}: {len(search)} search rows, {len(cluster)} cluster H rows")


def gen_lma(path, seed=77, per_scoring=40):
    """LinearMemoryAligner::align + alignstats outputs (the callers' fallback on the SHRT_MAX sentinel)"""
    rng = random.Random(seed)
    cases = []
    sets = [s for s in SCORINGS if s[0] in ("default", "distinct12", "nmismatch", "zero_terminal", "uniform10_1", "star_penalty")]
    for name, P, nmm in sets:
        ref = pyoracle.Reference(P, nmm)
        for _ in range(per_scoring):
            q, t = make_pair(rng, rng.choice(KINDS))
            cases.append({"scoring": name, "q": q, "t": t, "exp": list(ref.lma(q, t))})
        ref.close()
    doc = {"generator": "oracle/gen_golden.py (reference LinearMemoryAligner via oracle/_ref/libvsref.so)",
           "scorings": {n: {"P": list(P), "n_mismatch": nmm} for n, P, nmm in sets},
           "fields": ["score", "alnlen", "matches", "mismatches", "gaps", "cigar"], "cases": cases}
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(f"wrote {path}: {len(cases)} cases")


def dust_inputs(rng, n):
    """sequences with low-complexity stretches of every period, window-boundary lengths, some lower case and IUPAC codes"""
    def lowc(k):
        p = rng.choice([1, 2, 3, 4, 5, 7])
        u = "".join(rng.choice("ACGT") for _ in range(p))
        return (u * (k // p + 1))[:k]
    seqs = []
    edge = [1, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 95, 96, 97, 100, 127, 128, 129, 150, 250, 400, 1000]
    for i in range(n):
        L = edge[i] if i < len(edge) else rng.randint(1, 500)
        s = []
        while len(s) < L:
            if rng.random() < 0.3:
                s += list(lowc(rng.randint(4, 90)))
            else:
                s += [rng.choice("ACGT") for _ in range(rng.randint(1, 80))]
        s = s[:L]
        for j in range(len(s)):
            r = rng.random()
            if r < 0.02:
                s[j] = rng.choice("NRYKMSW")
            elif r < 0.06:
                s[j] = s[j].lower()
        seqs.append("".join(s))
    return seqs


def gen_dust(path, n=400):
    """DUST masking (core/mask.cpp) through the reference CLI: --fastx_mask --qmask dust"""
    import subprocess
    import tempfile
    from oracle import refcli
    assert refcli.available(), "oracle/_ref/vsearch_ref missing: make -C oracle ref_full"
    rng = random.Random(20)
    seqs = dust_inputs(rng, n)
    with tempfile.TemporaryDirectory(prefix="vsxdust_") as tmp:
        fa, out = os.path.join(tmp, "in.fa"), os.path.join(tmp, "out.fa")
        refcli.write_fasta(fa, [f"s{i}" for i in range(len(seqs))], seqs)
        subprocess.run([refcli.REF_BIN, "--fastx_mask", fa, "--qmask", "dust", "--fastaout", out, "--fasta_width", "0", "--quiet"], check=True)
        exp = [line.strip() for line in open(out) if not line.startswith(">")]
    assert len(exp) == len(seqs)
    doc = {"generator": "oracle/gen_golden.py gen_dust (reference CLI --fastx_mask --qmask dust)", "in": seqs, "exp": exp}
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(f"wrote {path}: {len(seqs)} sequences, {sum(1 for e in exp if any(c.islower() for c in e))} with a masked interval")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="write just this fixture: dust")
    ap.add_argument("--fuzz", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-write", action="store_true")
    a = ap.parse_args()
    pyoracle.build(ref=True)
    if a.only == "dust":
        gen_dust(os.path.join(GOLD, "dust_golden.json"))
        return
    if not a.no_write:
        os.makedirs(GOLD, exist_ok=True)
        gen_search16(os.path.join(GOLD, "search16_golden.json"))
        gen_api_examples(os.path.join(GOLD, "ref_api_examples.json"))
        gen_lma(os.path.join(GOLD, "lma_golden.json"))
        gen_dust(os.path.join(GOLD, "dust_golden.json"))
    if a.fuzz:
        sys.exit(1 if fuzz(a.fuzz, a.seed) else 0)


if __name__ == "__main__":
    main()
