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
