"""Pin the oracle to the only reference-produced numbers that exist for this path
(tests/golden/survey_known_answers.json, recorded in SURVEY.md section 8c)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "survey_known_answers.json")) as fh:
    KA = json.load(fh)


@pytest.mark.parametrize("case", KA["cases"], ids=lambda c: c["args"][-1])
def test_known_answers(oracle, case):
    east, north, up, east_goal, north_goal, up_goal, radius, aircraft, mission = case["args"]
    p = oracle.Problem(mission, aircraft, east_goal=east_goal, north_goal=north_goal, radius_goal=radius)
    assert (p.n, p.neF, p.neG) == (case["sizes"]["n"], case["sizes"]["neF"], case["sizes"]["neG"])
    x0 = p.x0()
    for entry in ("eval", "eval_entrywise"):
        F, G = getattr(p, entry)(x0)
        for idx, want in case["F"].items():
            # the survey printed 17 significant digits; the oracle evaluates the same formulas in a
            # different association order, so allow a few ulps
            assert F[int(idx)] == pytest.approx(want, rel=5e-15, abs=1e-13), (entry, idx)
        defined = ~p.undefined_mask()
        assert np.abs(G[defined]).sum() == pytest.approx(case["sum_abs_G_defined"], rel=1e-14)
        if "row4_slab_node0" in case:
            slab = G[p.c0 + 3 * 13: p.c0 + 4 * 13]
            assert slab == pytest.approx(np.array(case["row4_slab_node0"], dtype=float), rel=1e-11, abs=1e-12)


@pytest.mark.parametrize("row", KA["probed_sizes"], ids=lambda r: f"{r['mission']}-{r['N']}")
def test_probed_sizes(oracle, row):
    assert oracle.sizes(row["mission"], row["N"]) == (row["n"], row["neF"], row["neG"])
