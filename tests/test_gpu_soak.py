"""-m gpu: a short, seeded run of each randomized soak (oracle/soak.py, oracle/soak_search.py, oracle/soak_cluster.py, oracle/soak_allpairs.py): random
scoring sets / option sets / data against the reference's own search16 and the reference CLI.  The long runs (150 s each, other
seeds) are recorded in profiles/r02k_soak.json; they found three defects in round 2 (a TOPPAD eligibility hole for scoring sets
with ge(query left) > ge(query interior), minus-strand text missing for queries the device counters cannot serve, the order of
tied hits of both strands) -- each now also has its own regression test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,rounds,count_key,floor", [
    ("soak.py", 100, "pairs", 50_000),
    ("soak_search.py", 150, "userout_lines", 3_000),
    ("soak_cluster.py", 70, "uc_lines", 3_000),
    ("soak_allpairs.py", 150, "userout_lines", 5_000),
    ("soak_api.py", 25, "search_hits_compared", 100),
])
def test_seeded_soak(gpu_required, tmp_path, script, rounds, count_key, floor):
    """a FIXED set of rounds per script (seed + round count: the same configurations on every machine, whatever its speed)"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "vsearch_ref")) or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "api_driver_vsx")):
        pytest.fail("oracle/_ref missing: run `make -C oracle ref ref_full` in the build container")
    out = str(tmp_path / "soak.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", script), "--seconds", "300", "--max-rounds", str(rounds), "--seed", "20260924", "--out", out],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    doc = json.load(open(out))
    assert p.returncode == 0, json.dumps({k: v for k, v in doc.items() if k in ("mismatches", "failing_rounds", "failures", "examples")})[:3000]
    assert doc[count_key] >= floor, doc
