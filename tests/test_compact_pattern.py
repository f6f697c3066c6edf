"""The compact sparsity pattern: the reference pattern minus the
entries that are zero for every x.  CPU: the product's closed form equals the oracle's structural
selection and the dropped entries really are exact zeros.  GPU: values of the kept entries are the
same numbers, through the callback and the batch entry points."""
import numpy as np
import pytest

from helpers import assert_close, random_wind_table

AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("N", [1, 2, 7, 100, 200])
def test_compact_is_the_reference_pattern_minus_structural_zeros(tolfg, oracle, mission, N):
    o = oracle.Problem(mission, "tempest", N=N)
    idx = oracle.compact_index(o)
    fi, fj = o.pattern()
    pc = tolfg.Problem(mission, "tempest", ts=N, pattern="compact")
    ci, cj = pc.pattern()
    assert pc.neG == len(idx) == (49 * N + 26 if mission == "S10" else 47 * N + 36)
    assert np.array_equal(ci, fi[idx]) and np.array_equal(cj, fj[idx])
    assert (pc.n, pc.neF) == (o.n, o.neF)
    # x0, bounds do not depend on the pattern
    pf = tolfg.Problem(mission, "tempest", ts=N)
    assert np.array_equal(pc.x0(), pf.x0())
    pc.close(); pf.close()


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("aircraft", AIRCRAFT)
def test_dropped_entries_are_exact_zeros(oracle, mission, aircraft):
    """With a fully random wind Jacobian and random gains, every entry the compact pattern drops is
    exactly 0.0 in the reference-pattern G (they are never assigned, src/problem.cpp:1038)."""
    N = 50
    o = oracle.Problem(mission, aircraft, N=N, wind_table=random_wind_table(N, 9), gains=[1.3, 0.7, 0.4, 0.0, 0.9],
                       radius_goal=100.0 if mission == "S10" else 0.0)
    dropped = np.ones(o.neG, dtype=bool)
    dropped[oracle.compact_index(o)] = False
    assert dropped.sum() == 58 * N + o.nb
    for seed in range(5):
        G = o.eval(oracle.perturbed(o, seed))[1]
        assert (G[dropped] == 0.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("N", [1, 3, 52, 64, 65, 200])
@pytest.mark.parametrize("wind", ["shear", "table"])
def test_callback_compact_matches_oracle(tolfg, oracle, mission, N, wind):
    rg = 100.0 if mission == "S10" else 0.0
    table = random_wind_table(N, 4) if wind == "table" else None
    o = oracle.Problem(mission, "skywalker", N=N, radius_goal=rg, wind_table=table)
    p = tolfg.Problem(mission, "skywalker", ts=N, radius_goal=rg, pattern="compact")
    if table is not None:
        p.set_wind_table(table)
    idx = oracle.compact_index(o)
    for seed in (1, 2):
        x = oracle.perturbed(o, seed)
        F, G, st = p.define_fg(x)
        Fo, Go = o.eval(x)
        assert st == 1
        assert_close(F, Fo, what="compact F")
        assert_close(G, Go[idx], what="compact G")
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("pad", [None, 1])
def test_batch_compact_equals_reference_pattern_values(tolfg, oracle, mission, dtype, pad):
    """Same inputs through both patterns on the GPU: the kept entries are bitwise the same numbers."""
    import torch
    N, B = 200, 21
    rg = 100.0 if mission == "S10" else 0.0
    trajs = [tolfg.Trajectory(aircraft=t % 5, Vref=0.5 * t, href=8.0 + t, radius_goal=rg, xi=2.0 * t, yi=-t) for t in range(B)]
    out = {}
    for pat in ("reference", "compact"):
        bt = tolfg.Batch(mission, AIRCRAFT, ts=N, dtype=dtype, pattern=pat)
        bt.set_trajectories(trajs)
        if pat == "reference":
            ops = [oracle.Problem(mission, AIRCRAFT[tr.aircraft], N=N, radius_goal=rg, Vref=tr.Vref, href=tr.href,
                                  start=(tr.xi, tr.yi, -30.0)) for tr in trajs]
            X = np.stack([oracle.perturbed(ops[t], 300 + t) for t in range(B)])
        dX, dF, dG = bt.alloc(B, pad=pad)
        dX[:, :bt.n] = torch.from_numpy(X).to(bt.torch_dtype()).cuda()
        dG.fill_(float("nan"))
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        out[pat] = (dF[:, :bt.neF].cpu().numpy(), dG[:, :bt.neG].cpu().numpy(), bt.algorithmic_bytes(B))
        if dG.shape[1] > bt.neG:
            assert torch.isnan(dG[:, bt.neG:]).all()
    idx = oracle.compact_index(ops[0])
    assert np.array_equal(out["compact"][0], out["reference"][0])
    assert np.array_equal(out["compact"][1], out["reference"][1][:, idx])
    assert out["compact"][2] < 0.56 * out["reference"][2]
    if dtype == "f64":
        for t in range(B):
            assert_close(out["compact"][1][t], ops[t].eval(X[t])[1][idx], what=f"compact batch G[{t}]")
