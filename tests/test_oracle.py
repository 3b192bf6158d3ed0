"""CPU (-m "not gpu"): pin the oracle.  oracle/nw_oracle.c must reproduce
  (a) every fixture generated from the reference's own compiled sources (tests/golden/search16_golden.json,
      script oracle/gen_golden.py), and
  (b) the reference's in-tree golden files for this path (api_examples/data/expected_search.tsv,
      expected_cluster.uc; values carried in tests/golden/ref_api_examples.json),
and, where the reference tree is present (build container), agree with the reference on fresh random pairs."""
import random

import pytest

from tests import common
from tests.hostref import id_iddef2


def test_oracle_matches_reference_fixtures(oracle):
    doc = common.load_golden()
    bad = []
    for c in doc["cases"]:
        sc = doc["scorings"][c["scoring"]]
        got = oracle.align(c["q"], c["t"], sc["P"], sc["n_mismatch"])
        if list(got) != c["exp"]:
            bad.append((c["scoring"], c["q"][:20], c["t"][:20], c["exp"], got))
    assert not bad, f"{len(bad)} mismatches, first {bad[:2]}"
    assert len(doc["cases"]) > 800


def test_fixture_coverage():
    """the fixtures must actually exercise gaps, IUPAC, sentinels, empty sequences (edge cases the reference tests)"""
    doc = common.load_golden()
    cases = doc["cases"]
    assert any("I" in c["exp"][5] and "D" in c["exp"][5] for c in cases)
    assert any(c["exp"][0] == 32767 for c in cases)
    assert any(c["q"] == "" for c in cases) and any(c["t"] == "" for c in cases)
    assert any(set(c["q"]) - set("ACGT") for c in cases)
    assert {c["scoring"] for c in cases} == set(doc["scorings"])


def test_reference_in_tree_goldens(oracle):
    """expected_search.tsv: %id (iddef 2) of each reported (query, target) hit; expected_cluster.uc H rows:
    CIGAR + %id.  Values as printed by the reference (one decimal)."""
    ex = common.load_api_examples()
    seqs = dict(ex["refs"])
    seqs.update(ex["queries"])
    for row in ex["expected_search"]:
        r = oracle.align(seqs[row["query"]], ex["refs"][row["target"]])
        assert r[0] != 32767
        assert f"{id_iddef2(r[5], r[2]):.1f}" == row["id"], (row, r)
    for row in ex["expected_cluster_hits"]:
        r = oracle.align(ex["refs"][row["query"]], ex["refs"][row["target"]])
        assert r[5] == row["cigar"]
        assert f"{id_iddef2(r[5], r[2]):.1f}" == row["id"]


def test_closed_forms(oracle):
    """SURVEY.md Appendix A item 9, probed against the reference during the survey"""
    assert oracle.align("", "") == (0, 0, 0, 0, 0, "")
    assert oracle.align("ACGT", "") == (32767, 0, 0, 0, 0, "")
    assert oracle.align("", "ACGT") == (-5, 4, 0, 0, 4, "4I")
    assert oracle.align("", "A") == (-2, 1, 0, 0, 1, "1I")
    assert oracle.align("A" * 5001, "C" * 5000)[0] == 32767


def test_oracle_vs_live_reference():
    from oracle import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built here (no /root/reference): fixtures above still pin the oracle")
    orc = pyoracle.Oracle()
    rng = random.Random(4242)
    for P, nmm in [(pyoracle.DEFAULT_P, False), ((3, -5, 3, 7, 11, 13, 2, 5, 1, 2, 3, 4, 2, 1), True)]:
        ref = pyoracle.Reference(P, nmm)
        for _ in range(300):
            a = common.rnd_seq(rng, rng.randint(0, 150), "ACGTN")
            b = common.mutate(rng, a, 0.15) + common.rnd_seq(rng, rng.randint(0, 30))
            assert tuple(ref.align(a, b)) == tuple(orc.align(a, b, P, nmm)), (a, b)
        ref.close()


def test_lma_fallback_available_where_reference_is():
    """the sentinel contract: pairs search16 refuses are aligned by the caller's LinearMemoryAligner"""
    from oracle import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built here")
    ref = pyoracle.Reference()
    q, t = "ACGTACGTTTGACCA", "ACGTACGTTGACCA"
    sc, aln, ma, mi, ga, cig = ref.lma(q, t)
    assert (sc, aln, ma, mi, ga) == (8, 15, 14, 0, 1)
    ref.close()
