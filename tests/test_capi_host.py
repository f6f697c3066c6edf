"""The C-ABI library: it loads, exports every symbol include/tolfg.h declares, and its host-side
set-up (sizes, pattern, initial guess, bounds) equals the oracle's.  No GPU is needed for any of this;
evaluation without a GPU must fail loudly (there is no CPU path in the product)."""
import os
import re
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]


def declared_functions():
    text = open(os.path.join(ROOT, "include", "tolfg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(tolfg_[a-zA-Z0-9_]+|DEFINEGusrfg_)\s*\(", text))
    return names - {"tolfg_config", "tolfg_traj", "tolfg_batch_config"}


def test_every_declared_symbol_is_exported_and_bound(tolfg):
    from tol_amd import capi
    declared = declared_functions()
    assert "DEFINEGusrfg_" in declared and len(declared) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", tolfg.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert declared <= exported, declared - exported
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = tolfg.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.tolfg_version().startswith(b"tolfg")
    assert os.path.isdir(os.path.join(L.tolfg_default_root().decode(), "aircraft"))


def test_library_does_not_pin_a_hip_runtime(tolfg):
    """One HIP runtime per process: the library must not carry its own DT_NEEDED for libamdhip64
    (it would pull a second copy next to torch's and neither could open the GPU)."""
    from tol_amd import capi
    out = subprocess.run(["readelf", "-d", tolfg.lib_path()], capture_output=True, text=True, check=True).stdout
    assert "libamdhip64" not in out
    assert len(capi.mapped_hip_runtimes()) == 1


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("aircraft", AIRCRAFT)
@pytest.mark.parametrize("N", [0, 1, 7, 100, 200])
def test_setup_matches_oracle(tolfg, oracle, mission, aircraft, N):
    rg = 100.0 if mission == "S10" else 0.0
    p = tolfg.Problem(mission, aircraft, ts=N, east_goal=350.0, north_goal=40.0, radius_goal=rg, start=(3.0, -4.0, -20.0))
    o = oracle.Problem(mission, aircraft, N=(N or None), east_goal=350.0, north_goal=40.0, radius_goal=rg, start=(3.0, -4.0, -20.0))
    assert (p.n, p.neF, p.neG) == (o.n, o.neF, o.neG)
    for a, b in zip(p.pattern(), o.pattern()):
        assert np.array_equal(a, b)
    assert np.array_equal(p.x0(), o.x0())
    for a, b in zip(p.bounds(), o.bounds()):
        assert np.array_equal(a, b)
    assert p.tolerances() == (o.opt_tol, o.feas_tol)
    p.close()


def test_batch_setup_matches_oracle(tolfg, oracle):
    bt = tolfg.Batch("G7", AIRCRAFT, ts=33)
    trajs = [tolfg.Trajectory(aircraft=t % 5, north_goal=10.0 * t, east_goal=300.0 + t, radius_goal=0.0, xi=1.0 * t, yi=-2.0 * t)
             for t in range(7)]
    bt.set_trajectories(trajs)
    for t, tr in enumerate(trajs):
        o = oracle.Problem("G7", AIRCRAFT[tr.aircraft], N=33, east_goal=tr.east_goal, north_goal=tr.north_goal,
                           radius_goal=0.0, start=(tr.xi, tr.yi, -30.0))
        assert np.array_equal(bt.x0(t, zi=-30.0), o.x0())
        for a, b in zip(bt.bounds(t, zi=-30.0), o.bounds()):
            assert np.array_equal(a, b)
    for a, b in zip(bt.pattern(), o.pattern()):
        assert np.array_equal(a, b)
    assert bt.algorithmic_bytes(7) == 8 * 7 * (bt.n + bt.neF + bt.neG)


def test_bad_arguments(tolfg):
    with pytest.raises(tolfg.TolfgError):
        tolfg.Problem("Q1", "tempest")
    with pytest.raises(tolfg.TolfgError):
        tolfg.Batch("S10", [])
    with pytest.raises(tolfg.TolfgError):
        tolfg.Batch("S10", ["tempest"] * 9)
    bt = tolfg.Batch("S10", ["tempest"])
    with pytest.raises(tolfg.TolfgError):
        bt.set_trajectories([tolfg.Trajectory(aircraft=3)])


def test_no_current_problem_sets_status(tolfg):
    import ctypes as C
    L = tolfg.lib()
    L.tolfg_set_current(None)
    st, n, z = C.c_int(1), C.c_int(3), C.c_int(0)
    one = C.c_int(1)
    x = (C.c_double * 3)()
    L.DEFINEGusrfg_(C.byref(st), C.byref(n), x, C.byref(one), C.byref(n), x, C.byref(z), C.byref(n), x,
                    None, C.byref(z), None, C.byref(z), None, C.byref(z))
    assert st.value == -2


def test_evaluation_without_a_gpu_fails_loudly(tolfg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = tolfg.Problem("S10", "tempest")
    F, G, st = p.define_fg(p.x0())
    assert st == -2                     # snOptA: terminate; nothing was computed on the CPU
    assert (F == 0).all() and (G == 0).all()
    with pytest.raises(tolfg.TolfgError) as e:
        p.computeF(p.x0())
    assert e.value.code == -3


def test_batch_eval_argument_checks_need_no_gpu(tolfg):
    """Argument errors are reported before anything touches the device."""
    import ctypes as C
    L = tolfg.lib()
    bt = tolfg.Batch("S10", ["tempest"], ts=10)
    bt.set_trajectories([tolfg.Trajectory() for _ in range(3)])
    h = bt._h
    dummy = C.c_void_p(4096)           # never dereferenced: every call below must fail on its arguments

    def rc(B, dX, ldx, dF, ldf, dG, ldg, wind=None, needF=1, needG=1, obj=None):
        return L.tolfg_batch_eval(h, B, dX, ldx, dF, ldf, dG, ldg, wind, needF, needG, obj, None)

    n, neF, neG = bt.n, bt.neF, bt.neG
    assert rc(4, dummy, n, dummy, neF, dummy, neG) == -1                   # more trajectories than described
    assert rc(0, dummy, n, dummy, neF, dummy, neG) == -1
    assert rc(3, None, n, dummy, neF, dummy, neG) == -1                    # null X
    assert rc(3, dummy, n - 1, dummy, neF, dummy, neG) == -1               # row stride shorter than the row
    assert rc(3, dummy, n, dummy, neF, dummy, neG - 1) == -1
    assert rc(3, dummy, n, None, neF, dummy, neG) == -1                    # F wanted but null
    assert rc(3, dummy, n, None, neF, dummy, neG, needF=0, obj=dummy) == -1  # objectives need needF
    assert b"" != L.tolfg_last_error()
    bt2 = tolfg.Batch("S10", ["tempest"], ts=10, windmodel=99)
    bt2.set_trajectories([tolfg.Trajectory()])
    assert L.tolfg_batch_eval(bt2._h, 1, dummy, n, dummy, neF, dummy, neG, None, 1, 1, None, None) == -1   # table wind without a table
    assert L.tolfg_batch_x0_device(h, 3, None, n, None) == -1
    assert L.tolfg_write_json(None, None, 0.0, None) == -1


def test_python_layer_refuses_wrong_tensors_without_a_gpu(tolfg):
    """Batch.eval and friends check dtype, device, shape and strides before handing raw pointers to the C ABI
    (a float32 tensor given to an f64 batch would make the kernels write out of bounds)."""
    import torch
    bt = tolfg.Batch("S10", ["tempest"], ts=10)
    bt.set_trajectories([tolfg.Trajectory() for _ in range(3)])
    good = lambda c, dt=torch.float64: torch.zeros((3, c), dtype=dt)   # noqa: E731  (CPU tensors: wrong device)
    with pytest.raises(tolfg.TolfgError) as e:
        bt.eval(good(bt.n), good(bt.neF), good(bt.neG))
    assert e.value.code == tolfg.capi.ERR_ARG and "lives on" in str(e.value)
    with pytest.raises(tolfg.TolfgError) as e:
        bt.eval(good(bt.n, torch.float32), good(bt.neF), good(bt.neG))
    assert "dtype" in str(e.value)
    with pytest.raises(tolfg.TolfgError):
        bt.eval(np.zeros((3, bt.n)), good(bt.neF), good(bt.neG))
    with pytest.raises(tolfg.TolfgError):
        bt.x0_device(good(bt.n - 1))
    with pytest.raises(tolfg.TolfgError):
        bt.objectives(good(bt.neF))
    with pytest.raises(tolfg.TolfgError):
        bt.bounds_device(good(bt.n), good(bt.n), good(bt.neF), good(bt.neF))


def test_mixed_batch_set_up_needs_no_gpu(tolfg, oracle):
    """A "mixed" batch: per-mission sizes and patterns, per-trajectory x0 and bounds, argument checks."""
    bt = tolfg.Batch("mixed", ["tempest", "skywalker"], ts=20)
    nS, fS, gS = bt.sizes_of("S10")
    nG, fG, gG = bt.sizes_of("G7")
    assert (nS, fS, gS) == oracle.sizes("S10", 20) and (nG, fG, gG) == oracle.sizes("G7", 20)
    assert (bt.n, bt.neF, bt.neG) == (nS, max(fS, fG), max(gS, gG))
    for m in ("S10", "G7"):
        iG, jG = bt.pattern(m)
        oi, oj = oracle.Problem(m, N=20).pattern()
        assert np.array_equal(iG, oi) and np.array_equal(jG, oj)
    with pytest.raises(tolfg.TolfgError):
        bt.pattern()                                   # one pattern per mission
    trajs = [tolfg.Trajectory(aircraft=t % 2, mission=("S10", "G7")[t % 2], radius_goal=100.0 * (1 - t % 2), xi=3.0 * t, yi=-2.0 * t)
             for t in range(4)]
    bt.set_trajectories(trajs)
    for t, tr in enumerate(trajs):
        o = oracle.Problem(tr.mission, ("tempest", "skywalker")[t % 2], N=20, radius_goal=tr.radius_goal, start=(tr.xi, tr.yi, -30.0))
        n = bt.sizes_of(tr.mission)[0]
        assert np.array_equal(bt.x0(t, zi=-30.0)[:n], o.x0())
        for a, b in zip(bt.bounds(t, zi=-30.0), o.bounds()):
            assert np.array_equal(a[:len(b)], b)
    assert tolfg.lib().tolfg_batch_algorithmic_bytes(bt._h, 4) == 8.0 * (2 * (nS + fS + gS) + 2 * (nG + fG + gG))
    single = tolfg.Batch("S10", ["tempest"], ts=20)
    with pytest.raises(tolfg.TolfgError):
        single.sizes_of("G7")


def test_wind_model_codes_the_reference_leaves_empty_are_accepted_and_others_refused(tolfg):
    """The reference's arms 2, 4, 5 of modelWind have their bodies commented out (src/problem.cpp:534-542,698-730): the
    problem is built and evaluates without wind; a code the reference does not know is refused."""
    for wm in (2, 4, 5):
        p = tolfg.Problem("S10", "tempest", ts=20, windmodel=wm)
        assert (p.n, p.neF) == (232, 172)
        p.close()
        bt = tolfg.Batch("G7", ["tempest"], ts=20, windmodel=wm)
        bt.close()
    for wm in (6, 7, -1, 98):
        with pytest.raises(tolfg.TolfgError) as e:
            tolfg.Problem("S10", "tempest", ts=20, windmodel=wm)
        assert e.value.code == tolfg.capi.ERR_ARG


def test_the_shipped_library_holds_three_environment_variables_and_the_measurement_build_the_table(tolfg):
    """`strings libtolfg.so | grep -c TOLFG_` <= 4.  tol_amd/csrc/knobs.h tables every variable; the shipped
    library carries the three it documents in include/tolfg.h ("Environment"), the measurement build all of them."""
    import os
    import re
    import subprocess
    here = os.path.dirname(os.path.abspath(tolfg.capi.__file__))

    def names(path):
        out = subprocess.run(["strings", path], capture_output=True, text=True, check=True).stdout
        return sorted({m for ln in out.splitlines() for m in re.findall(r"TOLFG_[A-Z0-9_]+", ln)}), sum("TOLFG_" in ln for ln in out.splitlines())
    shipped, lines = names(os.path.join(here, "lib", "libtolfg.so"))
    assert shipped == ["TOLFG_MULTI_SHARED_DEVICES", "TOLFG_RCCL_LIBRARY", "TOLFG_TRACE"] and lines <= 4
    with open(os.path.join(here, "csrc", "knobs.h")) as fh:
        table = set(re.findall(r"^//\s+(TOLFG_[A-Z0-9_]+)", fh.read(), flags=re.M))
    measured, _ = names(os.path.join(here, "lib", "libtolfg_measure.so"))
    assert set(measured) == table and len(table) > 20
    with open(os.path.join(os.path.dirname(here), "include", "tolfg.h")) as fh:
        header = fh.read()
    assert all(v in header for v in shipped)
    assert tolfg.measure_lib().tolfg_measurement_build() == 1
    # no other file of the product calls getenv
    for f in os.listdir(os.path.join(here, "csrc")):
        if f.endswith((".cpp", ".hip", ".h")) and f != "knobs.cpp":
            with open(os.path.join(here, "csrc", f)) as fh:
                assert "getenv" not in fh.read(), f
