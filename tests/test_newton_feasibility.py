"""F, G and the exported pattern used the way an SQP driver uses them: Gauss-Newton on the equality constraints.

SNOPT is not available here, so this is the closest thing to "the existing driver calls it
unchanged and converges": starting from a perturbed initial guess, x <- x - J^+ c(x) with c = the rows whose bounds
are equal (the 8N defects and the periodicity rows), J assembled from (iGfun, jGvar, G) exactly as snOptA would.
With wind model 0 the reference's Jacobian is the true derivative (its frozen-wind approximation plays no part), so
the iteration must converge quadratically; a wrong entry, a wrong pattern position or an F / G mismatch shows up as a
stall.  CPU: the oracle.  GPU: DEFINEGusrfg_ through the C ABI, one callback per iteration.
"""
import numpy as np
import pytest


def gauss_newton(eval_fg, pattern, bounds, x, n, neF, iters=10):
    iG, jG = pattern
    _, _, Flow, Fupp = bounds
    rows = np.flatnonzero((Flow == Fupp) & (np.arange(neF) > 0))         # equalities; row 0 is the objective
    norms = []
    for _ in range(iters):
        F, G = eval_fg(x)
        c = F[rows] - Flow[rows]
        norms.append(np.abs(c).max())
        if norms[-1] < 1e-11:
            break
        J = np.zeros((neF, n))
        J[iG, jG] = G
        step, *_ = np.linalg.lstsq(J[rows], c, rcond=None)                # minimum-norm Newton step
        x = x - step
    return x, norms, len(rows)


def check_convergence(norms, what):
    assert norms[-1] < 1e-9, (what, norms)
    assert len(norms) <= 8, (what, norms)
    # quadratic tail: once below 1e-2 every iteration at least squares-ish the residual (allowing a constant of 50)
    late = [i for i, v in enumerate(norms[:-1]) if v < 1e-2]
    for i in late:
        assert norms[i + 1] <= max(50.0 * norms[i] ** 2, 1e-11), (what, norms)


@pytest.mark.parametrize("mission,N", [("S10", 40), ("G7", 30)])
def test_gauss_newton_on_the_oracle(oracle, mission, N):
    p = oracle.Problem(mission, "tempest", N=N, radius_goal=100.0 if mission == "S10" else 0.0, windmodel=oracle.WIND_NONE)
    rng = np.random.default_rng(5)
    x = p.x0()
    x = x + 0.01 * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))
    mask = p.undefined_mask()

    def eval_fg(v):
        F, G = p.eval(v)
        return F, np.where(mask, 0.0, G)
    x, norms, m = gauss_newton(eval_fg, p.pattern(), p.bounds(), x, p.n, p.neF)
    assert m == 8 * N + (11 if mission == "S10" else 11)          # G7: rows 9-19 are equalities, row 20 is dist <= dmax
    check_convergence(norms, f"oracle {mission}")


@pytest.mark.gpu
@pytest.mark.parametrize("mission,N,airframe", [("S10", 40, "tempest"), ("G7", 30, "skywalker"), ("S10", 200, "tempest")])
def test_gauss_newton_through_the_callback(tolfg, mission, N, airframe):
    p = tolfg.Problem(mission, airframe, radius_goal=100.0 if mission == "S10" else 0.0, ts=N, windmodel=tolfg.capi.WIND_NONE)
    rng = np.random.default_rng(6)
    x = p.x0()
    x = x + 0.01 * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))

    def eval_fg(v):
        F, G, st = p.define_fg(v)
        assert st == 1
        return F.copy(), G.copy()
    x, norms, _ = gauss_newton(eval_fg, p.pattern(), p.bounds(), x, p.n, p.neF)
    check_convergence(norms, f"callback {mission} ts={N}")
    p.close()


def test_frozen_wind_jacobian_converges_only_linearly(oracle):
    """The reference's Jacobian treats the wind as constant in position (dynamicsGradients), so under
    wind model 1 it is NOT the derivative of F: the same iteration then converges linearly (measured: one decade per step)
    instead of quadratically.  Reproduced as written -- this test documents the difference, it is not a defect of the build."""
    p = oracle.Problem("S10", "tempest", N=40, windmodel=oracle.WIND_SHEAR)
    rng = np.random.default_rng(5)
    x = p.x0()
    x = x + 0.01 * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))
    mask = p.undefined_mask()

    def eval_fg(v):
        F, G = p.eval(v)
        return F, np.where(mask, 0.0, G)
    _, norms, _ = gauss_newton(eval_fg, p.pattern(), p.bounds(), x, p.n, p.neF, iters=10)
    tail = [b / a for a, b in zip(norms[3:-1], norms[4:])]
    assert norms[-1] < 1e-8 and all(0.02 < r < 0.3 for r in tail), norms


@pytest.mark.gpu
def test_gauss_newton_on_a_batch(tolfg):
    """The batched evaluation in the same loop: eight trajectories (five air frames, different goals and starts), one launch
    per iteration for all of them, the linear algebra per trajectory on the host."""
    import torch
    N, B = 60, 8
    names = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]
    bt = tolfg.Batch("S10", names, ts=N, windmodel=tolfg.capi.WIND_NONE)
    bt.set_trajectories([tolfg.Trajectory(aircraft=t % 5, north_goal=10.0 * t, east_goal=400.0 - 20.0 * t, radius_goal=80.0 + 5.0 * t,
                                          xi=3.0 * t, yi=-2.0 * t, zi=-40.0 - t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    gen = torch.Generator(device="cuda").manual_seed(12)
    dX[:, :bt.n] += 0.01 * (torch.rand(B, bt.n, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1) * (1 + dX[:, :bt.n].abs())
    iG, jG = bt.pattern()
    dxl, dxu = torch.empty_like(dX), torch.empty_like(dX)
    dFl, dFu = torch.empty_like(dF), torch.empty_like(dF)
    bt.bounds_device(dxl, dxu, dFl, dFu)
    torch.cuda.synchronize()
    Fl, Fu = dFl.cpu().numpy(), dFu.cpu().numpy()
    rows = [np.flatnonzero((Fl[t, :bt.neF] == Fu[t, :bt.neF]) & (np.arange(bt.neF) > 0)) for t in range(B)]
    history = []
    for _ in range(8):
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        F, G, X = dF.cpu().numpy(), dG.cpu().numpy(), dX.cpu().numpy()
        worst = 0.0
        for t in range(B):
            c = F[t, rows[t]] - Fl[t, rows[t]]
            worst = max(worst, np.abs(c).max())
            J = np.zeros((bt.neF, bt.n))
            J[iG, jG] = G[t, :bt.neG]
            step, *_ = np.linalg.lstsq(J[rows[t]], c, rcond=None)
            X[t, :bt.n] -= step
        history.append(worst)
        if worst < 1e-11:
            break
        dX.copy_(torch.from_numpy(X))
    check_convergence(history, "batch")
    bt.close()
