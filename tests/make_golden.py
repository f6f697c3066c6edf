#!/usr/bin/env python3
"""Generate tests/golden/oracle_vectors.npz.

PROVENANCE: these vectors are produced by THIS REPO'S ORACLE (oracle/tolfg_oracle.c), not by the
reference -- the reference's hot path cannot be built in this image.  They
are a frozen copy of oracle outputs so that (a) a later change to the oracle that moves any number
is caught, and (b) the GPU box can check the HIP path against committed data.  The reference-produced
numbers are in tests/golden/survey_known_answers.json.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))             # lives under tests/: only test code may use oracle/
from oracle import oracle as O          # noqa: E402
from helpers import random_wind_table   # noqa: E402

CASES = [
    # name, mission, aircraft, N, wind ("none" | "shear" | "table"), seed
    ("S10_tempest_N100_shear", "S10", "tempest", 100, "shear", 7),
    ("S10_skywalker_N65_table", "S10", "skywalker", 65, "table", 8),
    ("S10_tempest_will_N3_none", "S10", "tempest_will", 3, "none", 9),
    ("G7_tempest_N100_shear", "G7", "tempest", 100, "shear", 7),
    ("G7_tempest_eric_N64_table", "G7", "tempest_eric", 64, "table", 8),
    ("G7_tempest_wences_N1_none", "G7", "tempest_wences", 1, "none", 9),
]


def build(case):
    name, mission, aircraft, N, wind, seed = case
    table = random_wind_table(N, 100 + seed) if wind == "table" else None
    p = O.Problem(mission, aircraft, N=N, radius_goal=100.0 if mission == "S10" else 0.0,
                  windmodel={"none": 0, "shear": 1, "table": 1}[wind], wind_table=table)
    x = O.perturbed(p, seed)
    F, G = p.eval(x)
    return p, table, x, F, G


def main():
    out = {}
    for case in CASES:
        p, table, x, F, G = build(case)
        out[case[0] + "/x"] = x
        out[case[0] + "/F"] = F
        out[case[0] + "/G"] = G
        if table is not None:
            out[case[0] + "/wind"] = table
    path = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
