"""Independent check of the oracle: central finite differences of its own F against its own G.

They must agree everywhere EXCEPT where the reference's Jacobian is knowingly not the derivative of
its F; those places are asserted too, so a silent 'fix' would also fail:
  quirk 2: wind is frozen in the Jacobian -> under shear wind d defect_x / d z is -dt*shear, G has 0
  quirk 3: G7 objective uses kv, its gradient kp
  quirk 4: G7 row 20 (dist - dmax) ignores dmax's dependence on (x0, y0)
"""
import numpy as np
import pytest

from helpers import random_wind_table


def dense_fd(p, x, h=1e-6):
    J = np.zeros((p.neF, p.n))
    for j in range(p.n):
        step = h * (1 + abs(x[j]))
        xp, xm = x.copy(), x.copy()
        xp[j] += step
        xm[j] -= step
        J[:, j] = (p.eval(xp, needG=False)[0] - p.eval(xm, needG=False)[0]) / (2 * step)
    return J


def dense_G(p, x):
    iG, jG = p.pattern()
    J = np.zeros((p.neF, p.n))
    J[iG, jG] = p.eval(x)[1]
    return J


def close(a, b, p=None, x=None):
    tol = 2e-6 * (1 + np.abs(b))
    if p is not None:
        # cancellation noise of the difference quotient grows with the size of the row's value
        # (the objective is ~1e5..1e7): eps * |F_i| / step
        Fabs = np.abs(p.eval(x, needG=False)[0])
        tol = tol + 1e-9 * Fabs[:, None]
    return np.abs(a - b) <= tol


@pytest.mark.parametrize("aircraft", ["tempest", "skywalker"])
def test_s10_table_wind_matches_fd_everywhere(oracle, aircraft):
    N = 6
    p = oracle.Problem("S10", aircraft, N=N, wind_table=random_wind_table(N, 5), gains=[0.7, 8.0, 0.0, 0.0, 1.0])
    x = oracle.perturbed(p, 11)
    assert close(dense_G(p, x), dense_fd(p, x), p, x).all()      # also: nothing non-zero outside the pattern


def test_g7_matches_fd_except_dmax(oracle):
    N = 6
    # kp == kv so that quirk 3 does not show; non-zero so that the distance terms are exercised
    p = oracle.Problem("G7", "tempest", N=N, radius_goal=0.0, wind_table=random_wind_table(N, 6),
                       gains=[100.0, 0.3, 0.3, 0.0, 0.0])
    x = oracle.perturbed(p, 12)
    ok = close(dense_G(p, x), dense_fd(p, x), p, x)
    bad = np.argwhere(~ok)
    row20 = p.neF - 1
    assert {tuple(b) for b in bad} == {(row20, 1), (row20, 2)}           # quirk 4, and nothing else


def test_g7_kp_kv_mismatch_is_reproduced(oracle):
    N = 5
    p = oracle.Problem("G7", "tempest", N=N, radius_goal=0.0, windmodel=0, gains=[100.0, 0.25, 0.75, 0.0, 0.0])
    x = oracle.perturbed(p, 13)
    G, FD = dense_G(p, x), dense_fd(p, x)
    cols = [0, 1, 2, 11 * N + 1, 11 * N + 2]
    # the value uses kv = 0.75, the gradient kp = 0.25: exactly a factor 3 on the distance terms
    assert np.allclose(FD[0, cols], 3.0 * G[0, cols], rtol=1e-3, atol=5e-6)   # FD noise ~ eps*|F0|/h


def test_shear_wind_is_frozen_in_the_jacobian(oracle):
    N = 6
    p = oracle.Problem("S10", "tempest", N=N, windmodel=1, Vref=2.4, href=10.0)
    x = oracle.perturbed(p, 14)
    G, FD = dense_G(p, x), dense_fd(p, x)
    bad = {tuple(b) for b in np.argwhere(~close(G, FD, p, x))}
    want = {(8 * k + 1, 11 * k + 3) for k in range(N)}                   # d defect_x(k) / d z_k
    assert bad == want
    for (i, j) in want:
        assert G[i, j] == 0.0
        assert FD[i, j] == pytest.approx(-x[0] * 0.24, rel=1e-5)


def test_s10_undefined_slots_are_zero_and_masked(oracle):
    p = oracle.Problem("S10", "tempest", N=9)
    m = p.undefined_mask()
    iG, jG = p.pattern()
    assert m.sum() == 11
    assert (jG[m] == 0).all() and (iG[m] >= 8 * 9 + 1).all()     # boundary rows x dt column
    assert (p.eval(oracle.perturbed(p, 1))[1][m] == 0.0).all()
