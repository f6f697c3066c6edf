"""The packed fp32 kernels: two collocation nodes per lane (v_pk_*_f32 arithmetic), tiles of up to 128 nodes.

The launch plan picks them for large mixed fp32 batches only (plan.cpp); here TOLFG_TILE_NODES=128 forces them onto small
batches so that every template instance meets the oracle: both missions and mixed batches, the four wind models, both
sparsity patterns, tiles that are full (128), ragged (one node in the second half, second half empty) and short, odd ts
(the slab stream's shifted forms) and a G that starts off a 16-byte boundary.  Bounds: the per-class fp32 bounds of
tests/helpers.py (the packed form has its own sin/cos and reciprocals: about 1 ulp each, like the library's)."""
import numpy as np
import pytest

from helpers import assert_close_f32, random_wind_table

pytestmark = pytest.mark.gpu

AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]


def _batch(tolfg, monkeypatch, mission, N, **kw):
    monkeypatch.setenv("TOLFG_TILE_NODES", "128")      # read by the measurement build only (tol_amd/csrc/knobs.h)
    return tolfg.Batch(mission, AIRCRAFT, ts=N, dtype="f32", library=tolfg.measure_lib(), **kw)


def _trajs(tolfg, mission, B):
    ms = [("S10", "G7")[t % 2] if mission == "mixed" else mission for t in range(B)]
    return ms, [tolfg.Trajectory(aircraft=t % 5, mission=ms[t], radius_goal=100.0 if ms[t] == "S10" else 0.0,
                                 Vref=0.5 + 0.4 * t, href=8.0 + t, xi=37.0 + 3.0 * t, yi=-41.0 - 2.0 * t) for t in range(B)]   # starts inside grid cells, not on their faces


def _oracles(oracle, ms, trajs, N, **kw):
    return [oracle.Problem(ms[t], AIRCRAFT[t % 5], N=N, radius_goal=trajs[t].radius_goal, Vref=trajs[t].Vref, href=trajs[t].href,
                           start=(trajs[t].xi, trajs[t].yi, -50.0), **kw) for t in range(len(trajs))]


@pytest.mark.parametrize("mission", ["S10", "G7", "mixed"])
@pytest.mark.parametrize("N", [65, 100, 128, 129, 200, 201, 300])
def test_packed_tiles_match_the_oracle(tolfg, oracle, monkeypatch, mission, N):
    import torch
    B = 7
    bt = _batch(tolfg, monkeypatch, mission, N)
    ms, trajs = _trajs(tolfg, mission, B)
    bt.set_trajectories(trajs)
    ops = _oracles(oracle, ms, trajs, N)
    X = np.stack([oracle.perturbed(ops[t], 300 + t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).to(torch.float32).cuda()
    dF.fill_(float("nan")); dG.fill_(float("nan"))
    for _ in range(2):                      # twice: the arrival counters and partial slots must be left ready
        bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    Xs = dX[:, :bt.n].double().cpu().numpy()
    for t in range(B):
        n, neF, neG = bt.sizes_of(ms[t]) if mission == "mixed" else (bt.n, bt.neF, bt.neG)
        iG = bt.pattern(ms[t])[0] if mission == "mixed" else bt.pattern()[0]
        Fo, Go = ops[t].eval(Xs[t, :n])
        Ft, Gt = dF[t, :neF].double().cpu().numpy(), dG[t, :neG].double().cpu().numpy()
        assert_close_f32(Ft, Gt, Fo, Go, iG, N, mask=ops[t].undefined_mask(), what=f"packed {mission} N={N} [{t}]")
        # nothing beyond the row's own sizes is written
        assert torch.isnan(dF[t, neF:]).all() and torch.isnan(dG[t, neG:]).all()


@pytest.mark.parametrize("wind", ["none", "table", "grid"])
@pytest.mark.parametrize("mission", ["S10", "G7"])
def test_packed_wind_models(tolfg, oracle, monkeypatch, mission, wind):
    import torch
    from test_wind_grid import make_grid
    N, B = 150, 5
    bt = _batch(tolfg, monkeypatch, mission, N, windmodel={"none": 0, "table": 99, "grid": 1}[wind])
    ms, trajs = _trajs(tolfg, mission, B)
    bt.set_trajectories(trajs)
    kw, scale = {}, 1.0
    dW = None
    if wind == "none":
        kw["windmodel"] = 0
    elif wind == "table":
        tables = [random_wind_table(N, 40 + t) for t in range(B)]
        dW = torch.from_numpy(np.stack(tables)).to(torch.float32).cuda()
    else:
        g = make_grid(9)
        bt.set_wind_grid(g["v"], g["origin"], g["spacing"], g["datum"])
        kw["wind_grid"] = dict(g, v=g["v"].astype(np.float32).astype(np.float64))
        scale = 4.0                          # cell coordinates formed in float32, as in tests/test_wind_grid.py
    ops = []
    for t in range(B):
        k = dict(kw)
        if wind == "table":
            k["wind_table"] = dW[t].double().cpu().numpy()
        ops.append(oracle.Problem(ms[t], AIRCRAFT[t % 5], N=N, radius_goal=trajs[t].radius_goal, Vref=trajs[t].Vref, href=trajs[t].href,
                                  start=(trajs[t].xi, trajs[t].yi, -50.0), **k))
    X = np.stack([oracle.perturbed(ops[t], 500 + t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).to(torch.float32).cuda()
    bt.eval(dX, dF, dG, wind=dW)
    torch.cuda.synchronize()
    Xs = dX[:, :bt.n].double().cpu().numpy()
    iG = bt.pattern()[0]
    for t in range(B):
        Fo, Go = ops[t].eval(Xs[t])
        assert_close_f32(dF[t, :bt.neF].double().cpu().numpy(), dG[t, :bt.neG].double().cpu().numpy(), Fo, Go, iG, N,
                         mask=ops[t].undefined_mask(), what=f"packed {mission}/{wind} [{t}]", scale=scale)


@pytest.mark.parametrize("mission", ["S10", "G7", "mixed"])
def test_packed_compact_pattern_and_shifted_rows(tolfg, oracle, monkeypatch, mission):
    """Compact 46-entry slabs (16-byte vectors that straddle nodes) and a G whose rows start 1, 2, 3 elements past a
    16-byte boundary; guard elements around every row must survive."""
    import torch
    N, B = 131, 6
    for pattern in ("reference", "compact"):
        bt = _batch(tolfg, monkeypatch, mission, N, pattern=pattern)
        ms, trajs = _trajs(tolfg, mission, B)
        bt.set_trajectories(trajs)
        ops = _oracles(oracle, ms, trajs, N)
        X = np.stack([oracle.perturbed(ops[t], 700 + t) for t in range(B)])
        dX, dF, _ = bt.alloc(B)
        dX[:, :bt.n] = torch.from_numpy(X).to(torch.float32).cuda()
        Xs = dX[:, :bt.n].double().cpu().numpy()
        ld = (bt.neG + 8 + 3) // 4 * 4
        for k in range(4):
            big = torch.full((B, ld), -777.0, dtype=torch.float32, device="cuda")
            dG = big[:, k:k + bt.neG + 1]
            bt.eval(dX, dF, dG)
            torch.cuda.synchronize()
            assert torch.all(big[:, :k] == -777.0), f"guard elements before the rows overwritten (offset {k})"
            for t in range(B):
                n, neF, neG = bt.sizes_of(ms[t]) if mission == "mixed" else (bt.n, bt.neF, bt.neG)
                assert torch.all(big[t, k + neG:] == -777.0), f"guard elements after row {t} overwritten (offset {k})"
                iG = bt.pattern(ms[t])[0] if mission == "mixed" else bt.pattern()[0]
                Fo, Go = ops[t].eval(Xs[t, :n])
                mask = ops[t].undefined_mask()
                if pattern == "compact":
                    idx = oracle.compact_index(ops[t])
                    Go, mask = Go[idx], mask[idx]
                assert_close_f32(dF[t, :neF].double().cpu().numpy(), big[t, k:k + neG].double().cpu().numpy(), Fo, Go, iG, N,
                                 mask=mask, what=f"packed {pattern} {mission} [{t}] offset {k}")


def test_packed_and_one_node_per_lane_agree(tolfg, monkeypatch):
    """Same inputs through both fp32 forms: different sin/cos and reciprocals, so close, not equal; and the packed form
    is really the one that ran (the two differ somewhere)."""
    import torch
    N, B = 200, 64
    out = {}
    for nodes in ("64", "128"):
        monkeypatch.setenv("TOLFG_TILE_NODES", nodes)
        bt = tolfg.Batch("mixed", AIRCRAFT, ts=N, dtype="f32", library=tolfg.measure_lib())
        _, trajs = _trajs(tolfg, "mixed", B)
        bt.set_trajectories(trajs)
        dX, dF, dG = bt.alloc(B)
        bt.x0_device(dX)
        dX[:, 1:bt.n] *= 1.01
        dF.zero_(); dG.zero_()
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        out[nodes] = (dF.double().cpu().numpy(), dG.double().cpu().numpy())
        bt.close()
    (F1, G1), (F2, G2) = out["64"], out["128"]
    assert np.abs(F1 - F2).max() > 0 or np.abs(G1 - G2).max() > 0
    assert (np.abs(G1 - G2) <= 4e-5 * (1 + np.abs(G1))).all()
    assert (np.abs(F1[:, 1:] - F2[:, 1:]) <= 4e-5 * (1 + np.abs(F1[:, 1:]))).all()
    assert (np.abs(F1[:, 0] - F2[:, 0]) <= 2e-6 * (1 + np.abs(F1[:, 0]))).all()


def test_the_shipped_library_ignores_the_measurement_variables(tolfg, monkeypatch):
    """tol_amd/csrc/knobs.h: the shipped libtolfg.so reads no variable that selects a kernel.  Under TOLFG_TILE_NODES=128 it runs
    what it runs without it -- bitwise what the measurement build runs when told to keep 64-node tiles -- while the
    measurement build under the same variable switches to the packed kernels (different sin/cos: different bits)."""
    import torch
    N, B = 200, 64
    _, trajs = _trajs(tolfg, "mixed", B)

    def run(library, nodes):
        if nodes is None:
            monkeypatch.delenv("TOLFG_TILE_NODES", raising=False)
        else:
            monkeypatch.setenv("TOLFG_TILE_NODES", nodes)
        bt = tolfg.Batch("mixed", AIRCRAFT, ts=N, dtype="f32", library=library)
        bt.set_trajectories(trajs)
        dX, dF, dG = bt.alloc(B)
        bt.x0_device(dX)
        dX[:, 1:bt.n] *= 1.01
        dF.zero_(); dG.zero_()
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        out = (dF.clone(), dG.clone())
        bt.close()
        return out
    if tolfg.lib().tolfg_measurement_build():
        pytest.skip("the suite runs against the measurement build (TOLFG_LIBRARY): nothing shipped to check")
    plain = run(tolfg.lib(), None)
    shipped = run(tolfg.lib(), "128")
    measured64 = run(tolfg.measure_lib(), "64")
    measured128 = run(tolfg.measure_lib(), "128")
    assert torch.equal(shipped[0], plain[0]) and torch.equal(shipped[1], plain[1])
    assert torch.equal(measured64[0], plain[0]) and torch.equal(measured64[1], plain[1])
    assert not torch.equal(measured128[1], plain[1])
