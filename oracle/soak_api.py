#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: randomized soak of the reference's LIBRARY API on the fast path (shim/vsx_api_adapter.cpp): every round
runs oracle/_ref/api_driver_vsx -- an embedder of src/vsearch_api.h -- with randomly drawn Parameters (the option sets of
oracle/soak_search.py) and data; inside, the reference's sequential search_session_single (its own code) is compared field by
field with search_batch (bound to libvsx), and every third round cluster_assign_single with cluster_assign_batch in ranges.

    python oracle/soak_api.py --seconds 100 --seed 1 --out gpurun_out/soak_api.json
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

from oracle import refcli, soak_cluster, soak_search  # noqa: E402

DRIVER = os.path.join(HERE, "_ref", "api_driver_vsx")
MASKS = ["none", "soft", "dust"]
SKIP = {"id", "maxaccepts", "maxrejects", "strand_both", "soft_mask", "qmask", "self_"}


def scoring_kv(scoring):
    if not scoring:
        return []
    e_e, e_i = scoring[8], scoring[10]
    return [f"match={scoring[0]}", f"mismatch={scoring[1]}", f"gapopen_e={scoring[2] + e_e}", f"gapopen_i={scoring[4] + e_i}", f"gapext_e={e_e}", f"gapext_i={e_i}"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--max-rounds", type=int, default=0, help="stop after this many rounds (0: run for --seconds): a deterministic set of rounds for a given seed")
    a = ap.parse_args()
    if not os.path.exists(DRIVER):
        raise SystemExit("oracle/_ref/api_driver_vsx missing: make -C oracle ref_full ref_api")
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    rounds = bad = hits = members = fast = 0
    failing = []
    env = dict(os.environ, VSX_ADAPTER_TRACE="1")
    with tempfile.TemporaryDirectory(prefix="vsxsoakapi_") as tmp:
        while time.time() < t_end and (a.max_rounds <= 0 or rounds < a.max_rounds):
            if rounds % 3 == 2:
                o, scoring, cli, by_size, round_size = soak_cluster.draw(rng)
                seqs, names, sz, order = soak_cluster.data(rng, bool(by_size))
                if by_size == "unoise":          # the library API has no unoise entry point: draw again
                    continue
                refcli.write_fasta(tmp + "/c.fa", names, seqs)         # the driver sorts by length itself (Database::sortbylength)
                kv = [f"{k}={v}" for k, v in o.items() if k not in ("id", "maxaccepts", "maxrejects", "soft_mask")] + scoring_kv(scoring)
                if by_size:
                    kv.append("sizes=1")
                argv = [DRIVER, "cluster", tmp + "/c.fa", repr(o["id"]), str(o["maxaccepts"]), str(o["maxrejects"]), str(rng.choice([5, 37, 100000])),
                        MASKS[o["soft_mask"]]] + kv
            else:
                o, scoring, cli, sizes, use_self = soak_search.draw_options(rng)
                db, qs, tn, qn, tsize, qsize = soak_search.draw_data(rng, sizes, use_self)
                refcli.write_fasta(tmp + "/db.fa", tn, db)
                refcli.write_fasta(tmp + "/q.fa", qn, qs)
                dbm = MASKS[o["soft_mask"]]
                qm = MASKS[o["qmask"] - 1] if o.get("qmask") else dbm
                kv = [f"{k}={v}" for k, v in o.items() if k not in SKIP] + scoring_kv(scoring)
                if sizes:
                    kv.append("sizes=1")
                if use_self:
                    kv.append("self=1")
                argv = [DRIVER, "search", tmp + "/db.fa", tmp + "/q.fa", repr(o["id"]), str(o["maxaccepts"]), str(o["maxrejects"]), str(o.get("strand_both", 0)),
                        qm, dbm] + kv
            p = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=env)
            rounds += 1
            fast += ("vsx_search_batch_meta" in p.stderr) or ("vsx_cluster_fast" in p.stderr)
            ok = p.returncode == 0 and p.stdout.strip().endswith(" 0 differences")
            if ok:
                if argv[1] == "search":
                    hits += int(p.stdout.split(" queries, ")[1].split(" hits")[0])
                else:
                    members += int(p.stdout.split(" clusters, ")[1].split(" members")[0])
            else:
                bad += 1
                if len(failing) < 10:
                    failing.append({"argv": argv[1:], "rc": p.returncode, "stdout": p.stdout[-300:], "stderr": p.stderr[-1200:], "round": rounds - 1})
    out = {"rounds": rounds, "rounds_on_the_fast_path": fast, "search_hits_compared": hits, "cluster_members_compared": members, "failing_rounds": bad,
           "failures": failing, "seed": a.seed, "seconds": a.seconds,
           "what": "api_driver_vsx: search_batch / cluster_assign_batch (shim/vsx_api_adapter.cpp -> libvsx) vs the reference's sequential entry points, every field"}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
