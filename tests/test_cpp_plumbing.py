"""BASELINE configs[0]: the SNOPT-side C++ plumbing.  A driver that includes only include/tolfg.h is
compiled with g++, linked `-ltolfg -lamdhip64` (what INTEGRATION.md tells a tol maintainer to do) and
enters DEFINEGusrfg_ through an snFunA pointer the way snoptProblemA::solve does."""
import os
import subprocess

import numpy as np
import pytest

from helpers import assert_close

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "cpp", "snopt_plumbing.cpp")
ROCM_LIB = "/opt/rocm/lib"


def build(tolfg, out):
    libdir = os.path.dirname(tolfg.lib_path())
    cmd = ["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L", libdir, "-ltolfg", "-L", ROCM_LIB, "-lamdhip64",
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{ROCM_LIB}"]
    subprocess.run(cmd, check=True)


def test_driver_compiles_and_links(tolfg, tmp_path):
    """CPU part: the snFunA assignment type-checks and the link line of INTEGRATION.md resolves."""
    exe = str(tmp_path / "plumbing")
    build(tolfg, exe)
    assert os.access(exe, os.X_OK)


@pytest.mark.gpu
@pytest.mark.parametrize("mission,aircraft,ts", [("G7", "tempest", 100), ("S10", "tempest", 100)])
def test_driver_runs_the_callback(tolfg, oracle, tmp_path, mission, aircraft, ts):
    exe = str(tmp_path / "plumbing")
    build(tolfg, exe)
    res = subprocess.run([exe, mission, aircraft, str(ts)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.splitlines()
    o = oracle.Problem(mission, aircraft, N=ts, radius_goal=100.0 if mission == "S10" else 0.0)
    assert lines[0] == f"status 1 n {o.n} neF {o.neF} neG {o.neG}"
    F = np.array([float(l[2:]) for l in lines if l.startswith("F ")])
    G = np.array([float(l[2:]) for l in lines if l.startswith("G ")])
    Fo, Go = o.eval(o.x0())
    assert_close(F, Fo, what="plumbing F")
    assert_close(G, Go, mask=o.undefined_mask(), what="plumbing G")
