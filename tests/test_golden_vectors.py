"""Committed vectors (tests/golden/oracle_vectors.npz, written by tests/make_golden.py from this
repo's oracle -- see that script for provenance): the oracle must still reproduce them bitwise on
CPU, and the HIP path must match them on the GPU box."""
import os
import sys

import numpy as np
import pytest

from helpers import assert_close

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import CASES   # noqa: E402

VEC = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))


def _oracle_problem(oracle, case):
    name, mission, aircraft, N, wind, seed = case
    table = VEC[name + "/wind"] if wind == "table" else None
    return oracle.Problem(mission, aircraft, N=N, radius_goal=100.0 if mission == "S10" else 0.0,
                          windmodel={"none": 0, "shear": 1, "table": 1}[wind], wind_table=table)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_oracle_reproduces_committed_vectors(oracle, case):
    p = _oracle_problem(oracle, case)
    F, G = p.eval(VEC[case[0] + "/x"])
    assert np.array_equal(F, VEC[case[0] + "/F"])
    assert np.array_equal(G, VEC[case[0] + "/G"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_hip_matches_committed_vectors(tolfg, oracle, case):
    name, mission, aircraft, N, wind, seed = case
    p = tolfg.Problem(mission, aircraft, ts=N, radius_goal=100.0 if mission == "S10" else 0.0,
                      windmodel={"none": 0, "shear": 1, "table": 1}[wind])
    if wind == "table":
        p.set_wind_table(VEC[name + "/wind"])
    F, G, st = p.define_fg(VEC[name + "/x"])
    assert st == 1
    mask = _oracle_problem(oracle, case).undefined_mask()
    assert_close(F, VEC[name + "/F"], what=name + " F")
    assert_close(G, VEC[name + "/G"], mask=mask, what=name + " G")
    p.close()
