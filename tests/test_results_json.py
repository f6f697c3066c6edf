"""`snopt_results.json`: same keys and values as the reference's writer
(ref: problem::writeJSON, src/problem.cpp:1247-1365), readable by its consumers
(msl/mission.py:208-226 reads trajectory/*, dt; matlab/@plotSNOPT reads the same)."""
import json

import numpy as np
import pytest

import os

# The key names the reference's writer assigns, extracted from its text at fixture time (tools/make_ref_vectors.py --set json-keys
# reads src/problem.cpp's writeJSON; names only are stored) -- not typed in here: a typo shared by this test and the product's
# writer would otherwise pass.
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "results_json_keys.json")) as _fh:
    _GOLDEN = json.load(_fh)
REF_KEYS = {k: set(v) for k, v in _GOLDEN["keys"].items()}


def test_the_golden_key_set_is_the_reference_writers():
    assert "writeJSON" in _GOLDEN["source"]
    assert set(REF_KEYS) == {"top", "args", "trajectory", "aircraft", "gains", "limits", "snopt"}
    assert REF_KEYS["top"] == {"args", "problem", "FinalCost", "dt", "trajectory", "aircraft", "gains", "limits", "snopt"}
    assert sum(len(v) for v in REF_KEYS.values()) == 65
    # what the reference's consumers read (msl/mission.py:208-226) is there
    assert {"time", "x", "y", "z"} <= REF_KEYS["trajectory"] and "dt" in REF_KEYS["top"]


@pytest.mark.parametrize("mission,aircraft", [("S10", "tempest"), ("G7", "skywalker")])
def test_results_json_schema_and_values(tolfg, oracle, tmp_path, mission, aircraft):
    N = 25
    p = tolfg.Problem(mission, aircraft, ts=N, east=1.0, north=2.0, up=100.0, east_goal=400.0, north_goal=5.0, up_goal=70.0,
                      radius_goal=90.0)
    o = oracle.Problem(mission, aircraft, N=N)
    x = oracle.perturbed(o, 2)
    out = tmp_path / "snopt_results.json"
    p.write_json(out, x, 1234.5678901234567)
    d = json.loads(out.read_text())
    assert set(d) == REF_KEYS["top"]
    for k in ("args", "trajectory", "aircraft", "gains", "limits", "snopt"):
        assert set(d[k]) == REF_KEYS[k], k
    assert d["problem"] == mission and d["args"]["problem"] == mission and d["aircraft"]["name"] == aircraft
    assert d["FinalCost"] == 1234.5678901234567 and d["dt"] == x[0]
    # goals are stored in NED like the reference's members: xg = north, yg = east, zg = -up
    assert (d["args"]["xg"], d["args"]["yg"], d["args"]["zg"], d["args"]["rd"]) == (5.0, 400.0, -70.0, 90.0)
    assert (d["args"]["east"], d["args"]["north"], d["args"]["up"]) == (1.0, 2.0, 100.0)
    node = x[1:].reshape(N + 1, 11)
    names = ["x", "y", "z", "Va", "gam", "chi", "phi", "CL", "dphi", "dCL", "T"]
    for m, nm in enumerate(names):
        assert np.array_equal(np.array(d["trajectory"][nm]), node[:, m]), nm      # every double round-trips
    t, tm = [], 0.0
    for _ in range(N + 1):
        t.append(tm)
        tm = tm + x[0]
    assert d["trajectory"]["time"] == t
    assert d["aircraft"]["mass"] == o.ac15[0] and d["aircraft"]["S"] == o.ac15[2]
    assert d["aircraft"]["phimax"] == o.ac15[8] * np.pi / 180.0
    assert d["gains"]["kp"] == o.gains[1] and d["snopt"]["ts"] == N and d["snopt"]["numbounds"] == o.nb
    assert d["limits"]["dtmax"] == o.lim8[1]
    p.close()
