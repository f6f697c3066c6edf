"""Initial guess and bounds generated on the device against the host
set-up, which equals the oracle bitwise (tests/test_capi_host.py)."""
import numpy as np
import pytest

from helpers import assert_close

pytestmark = pytest.mark.gpu
AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N", [1, 7, 200])
def test_device_x0_and_bounds(tolfg, oracle, mission, dtype, N):
    import torch
    B = 11
    rng = np.random.default_rng(5)
    rg = 100.0 if mission == "S10" else 0.0
    trajs = [tolfg.Trajectory(aircraft=t % 5, radius_goal=rg, north_goal=rng.uniform(-100, 100), east_goal=rng.uniform(200, 500),
                              xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50), zi=rng.uniform(-100, -20)) for t in range(B)]
    bt = tolfg.Batch(mission, AIRCRAFT, ts=N, dtype=dtype)
    bt.set_trajectories(trajs)
    dX, dF, dG = bt.alloc(B)
    dX.fill_(float("nan"))
    bt.x0_device(dX)
    xl, xu = torch.full_like(dX, float("nan")), torch.full_like(dX, float("nan"))
    Fl, Fu = torch.full_like(dF, float("nan")), torch.full_like(dF, float("nan"))
    bt.bounds_device(xl, xu, Fl, Fu)
    torch.cuda.synchronize()
    X = dX[:, :bt.n].double().cpu().numpy()
    tol = 1e-12 if dtype == "f64" else 1e-6
    cast = (lambda a: a) if dtype == "f64" else (lambda a: a.astype(np.float32).astype(np.float64))
    for t in range(B):
        want = bt.x0(t, zi=trajs[t].zi)
        assert_close(X[t], want, tol=tol, what=f"x0[{t}]")
        hl, hu, hFl, hFu = bt.bounds(t, zi=trajs[t].zi)
        assert np.array_equal(xl[t, :bt.n].double().cpu().numpy(), cast(hl))
        assert np.array_equal(xu[t, :bt.n].double().cpu().numpy(), cast(hu))
        assert np.array_equal(Fl[t, :bt.neF].double().cpu().numpy(), cast(hFl))
        assert np.array_equal(Fu[t, :bt.neF].double().cpu().numpy(), cast(hFu))
    if dX.shape[1] > bt.n:
        assert torch.isnan(dX[:, bt.n:]).all()
    # and the guess is a usable input: F, G of it match the oracle
    if dtype == "f64":
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        tr = trajs[3]
        o = oracle.Problem(mission, AIRCRAFT[tr.aircraft], N=N, east_goal=tr.east_goal, north_goal=tr.north_goal,
                           radius_goal=tr.radius_goal, start=(tr.xi, tr.yi, tr.zi))
        Fo, Go = o.eval(X[3])
        assert_close(dF[3, :bt.neF].cpu().numpy(), Fo, what="F(x0)")
        assert_close(dG[3, :bt.neG].cpu().numpy(), Go, mask=o.undefined_mask(), what="G(x0)")


@pytest.mark.parametrize("mission,dtype,N,B", [("S10", "f64", 200, 64), ("G7", "f64", 200, 64), ("mixed", "f64", 200, 130), ("mixed", "f32", 200, 66),
                                              ("S10", "f64", 255, 5), ("S10", "f64", 256, 5), ("G7", "f64", 257, 5), ("mixed", "f64", 2000, 6),
                                              ("S10", "f64", 1, 3), ("mixed", "f32", 700, 9)])
def test_node_parallel_x0_equals_the_serial_walk_bit_for_bit(tolfg, mission, dtype, N, B):
    """x0_kernel (one workgroup per trajectory, one thread per node) against x0_serial_kernel (one thread walks the nodes in
    order like the host code): the same rows, bit for bit -- courses that wrap (S10's lap crosses +-pi; G7 courses near
    +-pi), one pass and several passes of 256 nodes, node N's rates copied into node 0 (S10)."""
    import os
    import torch
    rng = np.random.default_rng(11)
    ms = [("S10", "G7")[t % 2] if mission == "mixed" else mission for t in range(B)]
    # goals all around the compass, so that G7's course chi_d = atan2(yg - yi, xg - xi) takes every sign and lies near +-pi too
    trajs = []
    for t in range(B):
        ang = -np.pi + 2 * np.pi * t / B + (1e-9 if t % 3 == 0 else 0.0)
        trajs.append(tolfg.Trajectory(aircraft=t % 5, mission=ms[t], radius_goal=100.0 if ms[t] == "S10" else 0.0,
                                      north_goal=300.0 * np.cos(ang), east_goal=300.0 * np.sin(ang),
                                      xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50), zi=rng.uniform(-100, -20)))
    bt = tolfg.Batch(mission, AIRCRAFT, ts=N, dtype=dtype)
    bt.set_trajectories(trajs)
    dXp, _, _ = bt.alloc(B)
    dXs = torch.full_like(dXp, float("nan"))
    dXp.fill_(float("nan"))
    bt.x0_device(dXp)
    # the serial reference form: a batch object of the measurement build created under TOLFG_X0_SERIAL (tol_amd/csrc/knobs.h)
    os.environ["TOLFG_X0_SERIAL"] = "1"
    try:
        bs = tolfg.Batch(mission, AIRCRAFT, ts=N, dtype=dtype, library=tolfg.measure_lib())
    finally:
        del os.environ["TOLFG_X0_SERIAL"]
    bs.set_trajectories(trajs)
    bs.x0_device(dXs)
    torch.cuda.synchronize()
    bs.close()
    a, b = dXp[:, :bt.n].cpu().numpy(), dXs[:, :bt.n].cpu().numpy()
    assert np.isfinite(b).all()
    bad = np.argwhere(a != b)
    assert bad.size == 0, f"{len(bad)} entries differ, first {bad[:5].tolist()}: {[(a[tuple(i)], b[tuple(i)]) for i in bad[:5]]}"
    # the lap really wraps: the continuous course of an S10 trajectory leaves (-pi, pi]
    if "S10" in ms and N >= 100:
        t = ms.index("S10")
        chi = dXp[t, 6:bt.n:11].double().cpu().numpy()
        assert np.abs(chi).max() > np.pi and np.abs(np.diff(chi)).max() < 1.0
    bt.close()
