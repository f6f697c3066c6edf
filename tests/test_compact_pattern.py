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


@pytest.mark.gpu
def test_batch_that_changes_launch_form_between_evaluations(tolfg, oracle):
    """The compact pattern takes the two-launch form when its outputs exceed the Infinity Cache (plan.cpp) and the
    single-launch form below; the two forms use the objective-partial slots differently (values left behind vs slots
    that must be "empty": batch::eval refills them when the form changes).  One batch object evaluated large, small, large,
    small with alternating inputs must give the right objectives each time.  (A slot left non-empty only matters when a
    partial's store is not yet visible to the finalizing wave, which this test cannot force; it guards the switching itself.)"""
    import torch
    N, B = 200, 2400                                   # 2400 x 91.5 KB = 220 MB of F + G: beyond the 192 MiB threshold
    bt = tolfg.Batch("S10", ["tempest", "skywalker"], ts=N, pattern="compact")
    bt.set_trajectories([tolfg.Trajectory(aircraft=t % 2, Vref=1.0 + 0.001 * t, radius_goal=90.0 + 0.01 * t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    gen = torch.Generator(device="cuda").manual_seed(21)
    dX[:, 1:bt.n] += 0.01 * torch.randn(B, bt.n - 1, dtype=torch.float64, device="cuda", generator=gen)
    # two different inputs alternate, so that a slot left over from the previous evaluation holds a WRONG value
    dX2 = dX.clone()
    dX2[:, 1::11] += 7.0                                # every node 7 m further north: the objective's sum of (r - R)^2 changes
    small = 48
    ref = {}
    for which, X in enumerate((dX, dX2)):
        for t in (0, 1, 17, small - 1, B - 1):
            o = oracle.Problem("S10", ["tempest", "skywalker"][t % 2], N=N, Vref=1.0 + 0.001 * t, radius_goal=90.0 + 0.01 * t)
            ref[(which, t)] = o.eval(X[t, :bt.n].cpu().numpy())[0]
    for i, Bnow in enumerate((B, small, B, small, small)):
        dF.fill_(float("nan"))
        bt.eval((dX, dX2)[i & 1], dF, dG, B=Bnow)
        torch.cuda.synchronize()
        F = dF.cpu().numpy()
        for (which, t), Fo in ref.items():
            if which == (i & 1) and t < Bnow:
                assert_close(F[t, :bt.neF], Fo, what=f"evaluation {i} B={Bnow} trajectory {t}")
    bt.close()
