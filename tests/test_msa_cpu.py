"""not gpu: the host star-MSA (vsx_msa) against the reference CLI on its own CIGARs.

oracle/_ref/vsearch_ref --cluster_fast writes --uc (member -> centroid CIGAR, the string the reference's msa() consumes)
next to --msaout / --consout / --profile; vsx_msa fed with those CIGARs must reproduce the three files byte for byte.
No GPU is involved: the clustering is the reference's, the MSA is host code."""
import os
import random
import subprocess

import pytest

from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vsearch_ref")


def _wrap(seq, width=80):
    return [seq[i:i + width] for i in range(0, len(seq), width)] or [""]


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/vsearch_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [3, 8])
def test_host_msa_matches_reference_cli(tmp_path, seed):
    from vsearch_amd import msa
    rng = random.Random(seed)
    seqs = []
    for f in range(6):
        anc = common.rnd_seq(rng, rng.randint(120, 180))
        for _ in range(rng.randint(1, 8)):
            seqs.append(common.mutate(rng, anc, rng.choice([0.02, 0.07])))
    seqs.append(common.mutate(rng, seqs[0], 0.03, "ACGTNRY"))
    seqs.append(seqs[1])                                   # an exact duplicate: the uc line carries '=' instead of a CIGAR
    rng.shuffle(seqs)
    names = [f"s{i:03d}" for i in range(len(seqs))]
    by_name = dict(zip(names, seqs))
    tmp = str(tmp_path)
    f_in = os.path.join(tmp, "m.fa")
    with open(f_in, "w") as f:
        f.write("".join(f">{n}\n{s}\n" for n, s in zip(names, seqs)))
    p = subprocess.run([REF_BIN, "--cluster_fast", f_in, "--id", "0.85", "--qmask", "none", "--threads", "1", "--quiet",
                        "--uc", tmp + "/m.uc", "--msaout", tmp + "/m.msa", "--consout", tmp + "/m.cons", "--profile", tmp + "/m.prof"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    clusters = {}                                          # cluster number -> [(name, cigar or None)], centroid first
    for ln in open(tmp + "/m.uc").read().splitlines():
        r = ln.split("\t")
        if r[0] == "S":
            clusters[int(r[1])] = [(r[8], None)]
        elif r[0] == "H":
            assert r[4] == "+"
            cig = r[7] if r[7] != "=" else f"{len(by_name[r[8]])}M"
            clusters[int(r[1])].append((r[8], cig))
    msa_lines, cons_lines, prof_lines = [], [], []
    for c in sorted(clusters):
        m = clusters[c]
        res = msa([by_name[n] for n, _ in m], [g for _, g in m])
        msa_lines.append("")
        for k, (n, _) in enumerate(m):
            msa_lines.append(">" + ("*" if k == 0 else "") + n)
            msa_lines += _wrap(res["rows"][k])
        msa_lines.append(">consensus")
        msa_lines += _wrap(res["rows"][-1])
        cons_lines.append(f">centroid={m[0][0]};seqs={len(m)}")
        cons_lines += _wrap(res["consensus"])
        prof_lines.append(f">centroid={m[0][0]};seqs={len(m)}")
        for i, (ch, pr) in enumerate(zip(res["rows"][-1], res["profile"])):
            prof_lines.append("\t".join([str(i), ch] + [str(pr[k]) for k in (0, 1, 2, 3, 5, 4)]))
        prof_lines.append("")
    assert msa_lines == open(tmp + "/m.msa").read().splitlines()
    assert cons_lines == open(tmp + "/m.cons").read().splitlines()
    assert prof_lines == open(tmp + "/m.prof").read().splitlines()
