"""Wind model 3 -- the gridded storm field with trilinear interpolation (SURVEY.md section 8f rank 2;
ref: problem::modelWind case 3, src/problem.cpp:544-695).  PARITY UNPINNED against the reference:
no storm data and no reference output exist offline.  What is checked: the oracle's restatement is
self-consistent (a linear field reproduces the table-wind path exactly; finite differences disagree
only where the reference freezes the wind), and the HIP path equals the oracle."""
import numpy as np
import pytest

from helpers import assert_close, assert_close_f32


def make_grid(seed, nx=7, ny=6, nz=4, linear=None):
    rng = np.random.default_rng(seed)
    origin = (17400.0 - 450.0, 25800.0 - 300.0, 200.0 - 300.0)   # around the reference's override point (:411-413)
    spacing = (150.0, 150.0, 150.0)
    datum = (17400.0, 25800.0, 200.0)
    if linear is not None:
        a, be, bn, bu = linear
        i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        v = a + be * (origin[0] + 150.0 * i) + bn * (origin[1] + 150.0 * j) + bu * (origin[2] + 150.0 * k)
    else:
        v = rng.uniform(-8, 8, (nx, ny, nz))
    return dict(v=v, origin=origin, spacing=spacing, datum=datum)


@pytest.mark.parametrize("mission", ["S10", "G7"])
def test_linear_field_equals_table_wind(oracle, mission):
    """Trilinear interpolation is exact on a field linear in (east, north, up); feeding the analytic
    v and gradients through the table-wind path must give the same F and G."""
    N = 30
    a, be, bn, bu = 1.5, 0.004, -0.003, 0.02
    rg = 100.0 if mission == "S10" else 0.0
    g = make_grid(0, linear=(a, be, bn, bu))
    og = oracle.Problem(mission, "tempest", N=N, radius_goal=rg, wind_grid=g)
    x = oracle.perturbed(og, 3)
    node = x[1:].reshape(N + 1, 11)
    east = node[:, 1] + g["datum"][0]
    north = node[:, 0] + g["datum"][1]
    up = -node[:, 2] + g["datum"][2]
    table = np.zeros((12, N + 1))
    table[1] = a + be * east + bn * north + bu * up     # v
    table[6], table[7], table[8] = be, bn, bu           # dv_dx(east), dv_dy(north), dv_dz(up)
    ot = oracle.Problem(mission, "tempest", N=N, radius_goal=rg, wind_table=table)
    Fg, Gg = og.eval(x)
    Ft, Gt = ot.eval(x)
    assert_close(Fg, Ft, tol=1e-11, what="grid vs table F")
    assert_close(Gg, Gt, tol=1e-11, what="grid vs table G")


def test_fd_disagrees_only_where_wind_is_frozen(oracle):
    from test_oracle_fd import dense_fd, dense_G, close
    N = 5
    o = oracle.Problem("S10", "tempest", N=N, wind_grid=make_grid(4), gains=[0.5, 8.0, 0.0, 0.0, 1.0])
    x = oracle.perturbed(o, 6)
    G, FD = dense_G(o, x), dense_fd(o, x, h=1e-7)
    bad = {tuple(b) for b in np.argwhere(~close(G, FD, o, x))}
    allowed = {(8 * k + r, 11 * k + 1 + m) for k in range(N) for r in (1, 4, 5, 6) for m in (0, 1, 2)}
    assert bad and bad <= allowed            # position columns of the rows the wind enters, nothing else


@pytest.mark.gpu
@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("N", [3, 64, 200])
def test_callback_grid_wind_matches_oracle(tolfg, oracle, mission, N):
    rg = 100.0 if mission == "S10" else 0.0
    g = make_grid(8)
    o = oracle.Problem(mission, "skywalker", N=N, radius_goal=rg, wind_grid=g)
    p = tolfg.Problem(mission, "skywalker", ts=N, radius_goal=rg)
    p.set_wind_grid(g["v"], g["origin"], g["spacing"], g["datum"])
    for seed in (1, 2):
        x = oracle.perturbed(o, seed)
        x[1:].reshape(N + 1, 11)[:, :2] *= 3.0          # spread over several cells, some nodes outside the grid
        F, G, st = p.define_fg(x)
        Fo, Go = o.eval(x)
        assert st == 1
        assert_close(F, Fo, what="grid F")
        assert_close(G, Go, mask=o.undefined_mask(), what="grid G")
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_batch_grid_wind_matches_oracle(tolfg, oracle, dtype):
    import torch
    N, B = 100, 13
    g = make_grid(9)
    bt = tolfg.Batch("S10", ["tempest", "skywalker"], ts=N, dtype=dtype)
    trajs = [tolfg.Trajectory(aircraft=t % 2, xi=20.0 * t, yi=-15.0 * t) for t in range(B)]
    bt.set_trajectories(trajs)
    bt.set_wind_grid(g["v"], g["origin"], g["spacing"], g["datum"])
    ops = [oracle.Problem("S10", ("tempest", "skywalker")[t % 2], N=N, start=(trajs[t].xi, trajs[t].yi, -50.0), wind_grid=g)
           for t in range(B)]
    X = np.stack([oracle.perturbed(ops[t], 70 + t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).to(bt.torch_dtype()).cuda()
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    Xs = dX[:, :bt.n].double().cpu().numpy()
    iG, _ = bt.pattern()
    for t in range(B):
        Fo, Go = ops[t].eval(Xs[t])
        if dtype == "f32":        # the float kernel interpolates a float32 copy of the grid
            g32 = dict(g, v=g["v"].astype(np.float32).astype(np.float64))
            Fo, Go = oracle.Problem("S10", ("tempest", "skywalker")[t % 2], N=N, start=(trajs[t].xi, trajs[t].yi, -50.0),
                                    wind_grid=g32).eval(Xs[t])
        Ft, Gt = dF[t, :bt.neF].double().cpu().numpy(), dG[t, :bt.neG].double().cpu().numpy()
        if dtype == "f64":
            assert_close(Ft, Fo, what=f"grid batch F[{t}]")
            assert_close(Gt, Go, mask=ops[t].undefined_mask(), what=f"grid batch G[{t}]")
        else:
            # the float kernel also forms the cell coordinates and shape functions in float32 from positions
            # of a few hundred metres: 4x the per-class bounds of the shear-wind sweep
            assert_close_f32(Ft, Gt, Fo, Go, iG, N, mask=ops[t].undefined_mask(), what=f"grid batch f32 [{t}]", scale=4.0)


def test_bad_grid_is_rejected(tolfg):
    bt = tolfg.Batch("S10", ["tempest"])
    with pytest.raises(tolfg.TolfgError):
        bt.set_wind_grid(np.zeros((1, 4, 4)), (0, 0, 0))
