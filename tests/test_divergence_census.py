"""The divergence census (tests/golden/divergence_census.json) decides which binary arithmetic is the by-size default:
the fused mode may only be a default if it leaves the decimal-15 (reference-semantics) pivot sequence no more often than
the plain mode, and if the 6-decimal result text (LPSolver.java:113) agrees wherever both reach an optimum.  This test
re-runs a sample of the census on the CPU oracle and checks that the committed file still says what the engine's policy
(choose_block in csrc/lpx_engine.cpp) relies on."""
import json
import os

import pytest

from tests.golden import gen_divergence_census as census

GOLD = os.path.join(os.path.dirname(__file__), "golden", "divergence_census.json")


@pytest.fixture(scope="module")
def doc():
    with open(GOLD) as f:
        return json.load(f)


def test_census_covers_what_the_verdict_asked_for(doc):
    recs = doc["records"]
    assert len(recs) >= 300
    fams = {r["family"] for r in recs}
    assert {"dense_u01", "dense_6dec", "packing_01", "degenerate_int"} <= fams
    sizes = {(r["m"], r["n"]) for r in recs if r["family"] == "dense_u01"}
    assert (64, 128) in sizes and (256, 512) in sizes
    assert any(r["phase1"] for r in recs)            # cfg5's family really runs phase 1


def test_fused_mode_is_no_less_faithful_than_plain(doc):
    """The policy's premise.  If this fails the fused mode must go back to opt-in at every size."""
    tot = doc["summary"]["TOTAL"]
    assert tot["fused"]["diverged"] <= tot["plain"]["diverged"]
    assert tot["fused"]["status_differs"] == 0 and tot["plain"]["status_differs"] == 0
    assert tot["fused"]["text_differs"] == 0 and tot["plain"]["text_differs"] == 0
    for r in doc["records"]:
        for mode in ("plain", "fused"):
            x = r[mode]
            if x["objective_rel_diff"] is not None:
                assert x["objective_rel_diff"] <= 1e-9, (r["family"], r["seed"], mode, x)
        if r["family"] != "packing_01":   # dense and dyadic-degenerate LPs: both binary modes walk the decimal pivots
            assert r["plain"]["first_divergence"] == -1 and r["fused"]["first_divergence"] == -1, r


def test_summary_is_the_summary_of_the_records(doc):
    assert census.summarise(doc["records"]) == doc["summary"]


def test_sampled_records_reproduce(doc):
    """Every 7th record of the small sizes, solved again by the three oracle instantiations."""
    small = [r for r in doc["records"] if r["m"] * r["n"] <= 96 * 160]
    sample = small[::7]
    assert len(sample) >= 30
    for r in sample:
        got = census.run_case((r["family"], r["m"], r["n"], r["seed"]))
        assert got == r, (r["family"], r["m"], r["n"], r["seed"])
