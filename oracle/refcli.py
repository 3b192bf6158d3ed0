"""TEST / BENCH INFRASTRUCTURE ONLY: thin runner of the reference CLI (oracle/_ref/vsearch_ref = the reference's own sources
compiled in place by oracle/Makefile ref_full).  Used by the secondary benches to cross-check a SAMPLE of their full-size
workload against the reference (`parity_sample_match`); never on the product path."""
import os
import subprocess
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "vsearch_ref")


def available():
    return os.path.exists(REF_BIN)


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def write_fasta(path, names, seqs):
    with open(path, "wb") as f:
        parts = []
        for n, s in zip(names, seqs):
            parts.append(b">" + n.encode() + b"\n" + (s if isinstance(s, bytes) else s.encode()) + b"\n")
            if len(parts) >= 65536:
                f.write(b"".join(parts))
                parts = []
        f.write(b"".join(parts))


def run(args):
    t0 = time.time()
    p = subprocess.run([REF_BIN] + args, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(p.stderr[-2000:])
    return time.time() - t0


def allpairs_userout(names, seqs, idv, fields=("query", "target", "id", "caln")):
    """--allpairs_global --userout lines (sorted) and the wall time of the command"""
    with tempfile.TemporaryDirectory(prefix="vsxref_") as tmp:
        fa, uo = os.path.join(tmp, "a.fa"), os.path.join(tmp, "u.tsv")
        write_fasta(fa, names, seqs)
        secs = run(["--allpairs_global", fa, "--id", repr(idv), "--qmask", "none", "--threads", str(usable_cpus()),
                    "--userout", uo, "--userfields", "+".join(fields), "--quiet"])
        return sorted(open(uo).read().splitlines()), secs


def cluster_fast_uc(names, seqs, idv):
    """--cluster_fast --uc lines and the wall time of the command"""
    with tempfile.TemporaryDirectory(prefix="vsxref_") as tmp:
        fa, uc = os.path.join(tmp, "c.fa"), os.path.join(tmp, "c.uc")
        write_fasta(fa, names, seqs)
        secs = run(["--cluster_fast", fa, "--id", repr(idv), "--qmask", "none", "--threads", str(usable_cpus()), "--uc", uc, "--quiet"])
        return open(uc).read().splitlines(), secs
