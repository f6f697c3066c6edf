"""The closed-form sparsity pattern against a restatement of countG's dense walk
(ref: src/problem.cpp:813-919), and the two evaluation orders against each other."""
import numpy as np
import pytest


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("N", [1, 2, 3, 4, 9, 33, 100])
def test_closed_form_equals_dense_walk(oracle, mission, N):
    p = oracle.Problem(mission, "tempest", N=N)
    iG, jG = p.pattern()
    iW, jW = p.pattern(walk=True)
    assert len(iW) == p.neG
    assert np.array_equal(iG, iW) and np.array_equal(jG, jW)
    # row-major sorted, no duplicates: the order countG's double loop produces
    key = iG.astype(np.int64) * p.n + jG
    assert (np.diff(key) > 0).all()
    for a, b in zip(p.dispatch(), p.dispatch(walk=True)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mission", ["S10", "G7"])
def test_slab_layout(oracle, mission):
    """Node k owns G[c0+104k, +104) = 8 rows x [dt | 11 node vars | next-node state]."""
    N = 7
    p = oracle.Problem(mission, "tempest", N=N)
    iG, jG = p.pattern()
    for k in range(N):
        sl = slice(p.c0 + 104 * k, p.c0 + 104 * (k + 1))
        rows = iG[sl].reshape(8, 13)
        cols = jG[sl].reshape(8, 13)
        for r in range(8):
            assert (rows[r] == 8 * k + r + 1).all()
            assert cols[r, 0] == 0
            assert np.array_equal(cols[r, 1:12], 11 * k + 1 + np.arange(11))
            assert cols[r, 12] == 11 * (k + 1) + r + 1


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("wind", [0, 1, 99])
def test_entrywise_equals_fused(oracle, mission, wind):
    """Reference evaluation order (one entry per call) and the per-node fused order agree bitwise."""
    from helpers import random_wind_table
    N = 40
    table = random_wind_table(N, 3) if wind == 99 else None
    p = oracle.Problem(mission, "skywalker", N=N, windmodel=min(wind, 1), wind_table=table)
    for seed in (1, 2):
        x = oracle.perturbed(p, seed)
        F1, G1 = p.eval(x)
        F2, G2 = p.eval_entrywise(x)
        assert np.array_equal(F1, F2) and np.array_equal(G1, G2)
        F3, G3 = p.eval(x, opt="O0")           # -O0 and -O2 builds agree bitwise (no contraction)
        assert np.array_equal(F1, F3) and np.array_equal(G1, G3)
