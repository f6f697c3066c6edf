"""HIP path vs oracle, through the C ABI (DEFINEGusrfg_ and the batched entry points)."""
import numpy as np
import pytest

from helpers import assert_close, assert_close_f32, random_wind_table

pytestmark = pytest.mark.gpu

AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]


def _goal(mission):
    return dict(east_goal=400.0, north_goal=0.0, radius_goal=100.0 if mission == "S10" else 0.0)


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("N", [1, 2, 3, 63, 64, 65, 100, 127, 128, 200, 333])
def test_callback_matches_oracle(tolfg, oracle, mission, N):
    """DEFINEGusrfg_ with SNOPT's calling convention at x0 and at seeded perturbed points."""
    p = tolfg.Problem(mission, "tempest", ts=N, **_goal(mission))
    o = oracle.Problem(mission, "tempest", N=N, **_goal(mission))
    mask = o.undefined_mask()
    pts = [o.x0()] + [oracle.perturbed(o, seed) for seed in (7, 8, 9)]
    for i, x in enumerate(pts):
        F, G, st = p.define_fg(x)
        assert st == 1
        Fo, Go = o.eval(x)
        assert_close(F, Fo, what=f"{mission} N={N} pt{i} F")
        assert_close(G, Go, mask=mask, what=f"{mission} N={N} pt{i} G")
        assert (G[mask] == 0.0).all()      # the reference's undefined slots are defined as 0 here
    p.close()


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("aircraft", AIRCRAFT)
@pytest.mark.parametrize("wind", ["none", "shear", "table"])
def test_airframes_and_wind_models(tolfg, oracle, mission, aircraft, wind):
    N = 100
    wm = {"none": 0, "shear": 1, "table": 1}[wind]
    p = tolfg.Problem(mission, aircraft, ts=N, windmodel=wm, Vref=3.1, href=12.5, **_goal(mission))
    table = random_wind_table(N, 11) if wind == "table" else None
    o = oracle.Problem(mission, aircraft, N=N, windmodel=wm, Vref=3.1, href=12.5, wind_table=table, **_goal(mission))
    if table is not None:
        p.set_wind_table(table)
    mask = o.undefined_mask()
    for seed in (21, 22):
        x = oracle.perturbed(o, seed)
        F, G, st = p.define_fg(x)
        assert st == 1
        Fo, Go = o.eval(x)
        assert_close(F, Fo, what=f"{mission}/{aircraft}/{wind} F")
        assert_close(G, Go, mask=mask, what=f"{mission}/{aircraft}/{wind} G")
    p.close()


@pytest.mark.parametrize("code", [2, 4, 5])
def test_wind_model_codes_whose_reference_arms_are_empty_evaluate_without_wind(tolfg, oracle, code):
    """src/problem.cpp:534-542,698-730: the thermal / two-thermal / cyclic arms are commented out, the wind vectors stay zero."""
    p = tolfg.Problem("S10", "tempest", ts=60, windmodel=code, **_goal("S10"))
    o = oracle.Problem("S10", "tempest", N=60, windmodel=0, **_goal("S10"))
    x = oracle.perturbed(o, 31)
    F, G, st = p.define_fg(x)
    Fo, Go = o.eval(x)
    assert st == 1
    assert_close(F, Fo, what="F")
    assert_close(G, Go, mask=o.undefined_mask(), what="G")
    p.close()


def test_need_flags_and_untouched_outputs(tolfg, oracle):
    """needF/needG = 0 must leave the corresponding array untouched (src/DefineFG.cpp:26,36)."""
    p = tolfg.Problem("S10", "tempest", ts=100, **_goal("S10"))
    o = oracle.Problem("S10", "tempest", N=100, **_goal("S10"))
    x = oracle.perturbed(o, 5)
    Fo, Go = o.eval(x)
    F, G, st = p.define_fg(x, needF=True, needG=False)
    assert_close(F, Fo, what="F only")
    assert (G == 0).all()
    F, G, st = p.define_fg(x, needF=False, needG=True)
    assert (F == 0).all()
    assert_close(G, Go, mask=o.undefined_mask(), what="G only")
    F, G, st = p.define_fg(x, needF=False, needG=False)
    assert (F == 0).all() and (G == 0).all() and st == 1
    p.close()


def test_three_methods_and_iu_route(tolfg, oracle):
    """modelWind/computeF/computeG driven separately, and two problems alive at once via iu[]."""
    a = tolfg.Problem("S10", "tempest", ts=100, **_goal("S10"))
    b = tolfg.Problem("G7", "skywalker", ts=64, **_goal("G7"))
    oa = oracle.Problem("S10", "tempest", N=100, **_goal("S10"))
    ob = oracle.Problem("G7", "skywalker", N=64, **_goal("G7"))
    xa, xb = oracle.perturbed(oa, 1), oracle.perturbed(ob, 2)
    a.modelWind(xa)
    assert_close(a.computeF(xa), oa.eval(xa)[0], what="computeF")
    assert_close(a.computeG(xa), oa.eval(xa)[1], mask=oa.undefined_mask(), what="computeG")
    xa2 = oracle.perturbed(oa, 3)           # computeG with an x that was never staged
    assert_close(a.computeG(xa2), oa.eval(xa2)[1], mask=oa.undefined_mask(), what="computeG restage")
    a.make_current()
    Fb, Gb, st = b.define_fg(xb, use_iu=True)   # current is `a`, iu[] selects `b`
    assert st == 1
    assert_close(Fb, ob.eval(xb)[0], what="iu F")
    assert_close(Gb, ob.eval(xb)[1], what="iu G")
    a.close(); b.close()


def test_size_mismatch_sets_status(tolfg):
    import ctypes as C
    p = tolfg.Problem("S10", "tempest", ts=100, **_goal("S10"))
    x = p.x0()[:-1].copy()
    F, G, st = p.define_fg(x)
    assert st == -2
    p.close()


@pytest.mark.parametrize("mission", ["S10", "G7"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("pad", [None, 1])
def test_batch_matches_oracle(tolfg, oracle, mission, dtype, pad):
    """Mixed airframes, per-trajectory shear wind and goals, ragged B, aligned and unaligned rows."""
    import torch
    N, B = 200, 37
    rng = np.random.default_rng(123)
    bt = tolfg.Batch(mission, AIRCRAFT, ts=N, dtype=dtype)
    trajs, oprobs, X = [], [], []
    for t in range(B):
        tr = tolfg.Trajectory(aircraft=t % 5, Vref=rng.uniform(0, 5), href=rng.uniform(5, 20),
                              north_goal=rng.uniform(-50, 50), east_goal=rng.uniform(300, 500),
                              radius_goal=(rng.uniform(50, 150) if mission == "S10" else 0.0),
                              xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50))
        trajs.append(tr)
        o = oracle.Problem(mission, AIRCRAFT[tr.aircraft], N=N, east_goal=tr.east_goal, north_goal=tr.north_goal,
                           radius_goal=tr.radius_goal, start=(tr.xi, tr.yi, -40.0), Vref=tr.Vref, href=tr.href)
        oprobs.append(o)
        X.append(oracle.perturbed(o, 1000 + t))
    bt.set_trajectories(trajs)
    # x0 of the batch API agrees with the oracle's
    assert np.array_equal(bt.x0(3, zi=-40.0), oprobs[3].x0())
    X = np.stack(X)
    dX, dF, dG = bt.alloc(B, pad=pad)
    tdt = bt.torch_dtype()
    dX[:, :bt.n] = torch.from_numpy(X).to(tdt).cuda()
    dF.fill_(float("nan")); dG.fill_(float("nan"))
    obj2 = torch.full((B,), float("nan"), dtype=tdt, device=dX.device)
    bt.eval(dX, dF, dG, obj=obj2)
    obj = bt.objectives(dF)
    torch.cuda.synchronize()
    assert torch.equal(obj, obj2)                 # fused objective output == separate gather kernel
    F = dF[:, :bt.neF].double().cpu().numpy()
    G = dG[:, :bt.neG].double().cpu().numpy()
    # padding columns must not be written
    if dF.shape[1] > bt.neF:
        assert torch.isnan(dF[:, bt.neF:]).all()
    if dG.shape[1] > bt.neG:
        assert torch.isnan(dG[:, bt.neG:]).all()
    Xin = dX[:, :bt.n].double().cpu().numpy()       # what the kernel really saw (rounded for f32)
    iG, _ = bt.pattern()
    for t in range(B):
        Fo, Go = oprobs[t].eval(Xin[t])
        if dtype == "f64":
            assert_close(F[t], Fo, what=f"batch {mission} f64 F[{t}]")
            assert_close(G[t], Go, mask=oprobs[t].undefined_mask(), what=f"batch {mission} f64 G[{t}]")
        else:       # per row class, at the measured fp32 accuracy (helpers.FP32_TOL)
            assert_close_f32(F[t], G[t], Fo, Go, iG, N, mask=oprobs[t].undefined_mask(), what=f"batch {mission} f32 [{t}]")
    assert_close(obj.double().cpu().numpy(), F[:, 0], tol=0.0, what="objectives gather")


def test_eval_is_graph_capturable(tolfg, oracle):
    """After one warm call (workspace allocation, trajectory upload) tolfg_batch_eval only enqueues
    kernels, so a caller may capture it into a hipGraph and replay it (launch-bound inner loops)."""
    import torch
    N, B = 200, 16
    bt = tolfg.Batch("S10", ["tempest"], ts=N)
    bt.set_trajectories([tolfg.Trajectory(Vref=1.0 + t) for t in range(B)])
    ops = [oracle.Problem("S10", "tempest", N=N, Vref=1.0 + t) for t in range(B)]
    X = np.stack([oracle.perturbed(ops[t], 500 + t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    obj = torch.zeros(B, dtype=torch.float64, device="cuda")
    dX[:, :bt.n] = torch.from_numpy(X).cuda()
    bt.eval(dX, dF, dG, obj=obj)                      # warm call outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        bt.eval(dX, dF, dG, obj=obj)
    X2 = np.stack([oracle.perturbed(ops[t], 900 + t) for t in range(B)])
    dX[:, :bt.n] = torch.from_numpy(X2).cuda()        # new inputs, same buffers
    dF.zero_(); dG.zero_(); obj.zero_()
    g.replay()
    torch.cuda.synchronize()
    for t in (0, 7, 15):
        Fo, Go = ops[t].eval(X2[t])
        assert_close(dF[t, :bt.neF].cpu().numpy(), Fo, what="graph F")
        assert_close(dG[t, :bt.neG].cpu().numpy(), Go, mask=ops[t].undefined_mask(), what="graph G")
    assert np.array_equal(obj.cpu().numpy(), dF[:, 0].cpu().numpy())


def test_unguarded_divisions_propagate_like_the_reference(tolfg, oracle):
    """cost() divides by r_k (S10) / dist (G7) unguarded, so a node on
    the goal centre or a closed G7 leg yields inf/nan.  The HIP path must produce non-finite values
    in exactly the entries where the oracle does, and agree everywhere else."""
    N = 64
    # S10: put node 10 exactly on the goal centre -> r = 0 -> 0/0 in its two objective-gradient entries
    o = oracle.Problem("S10", "tempest", N=N, east_goal=400.0, north_goal=0.0)
    p = tolfg.Problem("S10", "tempest", ts=N, east_goal=400.0, north_goal=0.0)
    x = oracle.perturbed(o, 4)
    x[11 * 10 + 1], x[11 * 10 + 2] = 0.0, 400.0
    F, G, st = p.define_fg(x)
    Fo, Go = o.eval(x)
    assert st == 1
    assert np.array_equal(np.isfinite(G), np.isfinite(Go)) and np.array_equal(np.isfinite(F), np.isfinite(Fo))
    bad = ~np.isfinite(Go)
    assert bad.sum() == 2 and set(np.flatnonzero(bad)) == {1 + 3 * 10, 2 + 3 * 10}
    assert_close(np.where(bad, 0.0, G), np.where(bad, 0.0, Go), mask=o.undefined_mask(), what="S10 r=0 G")
    assert_close(F, Fo, what="S10 r=0 F")
    p.close()
    # G7: last node on top of the first -> dist = 0
    o = oracle.Problem("G7", "tempest", N=N, radius_goal=0.0, gains=[100.0, 0.5, 0.5, 0.0, 0.0])
    p = tolfg.Problem("G7", "tempest", ts=N, radius_goal=0.0)
    x = oracle.perturbed(o, 5)
    x[11 * N + 1], x[11 * N + 2] = x[1], x[2]
    F, G, st = p.define_fg(x)
    o_ship = oracle.Problem("G7", "tempest", N=N, radius_goal=0.0)     # shipped gains (kp = kv = 0), like the product
    Fo, Go = o_ship.eval(x)
    assert st == 1
    assert np.array_equal(np.isnan(G), np.isnan(Go)) and np.array_equal(np.isinf(G), np.isinf(Go))
    assert np.array_equal(np.isnan(F), np.isnan(Fo)) and np.array_equal(np.isinf(F), np.isinf(Fo))
    assert (~np.isfinite(Go)).sum() > 0
    ok = np.isfinite(Go)
    assert_close(np.where(ok, G, 0.0), np.where(ok, Go, 0.0), what="G7 dist=0 G")
    p.close()


@pytest.mark.parametrize("pattern", ["reference", "compact"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mission", ["S10", "G7"])
def test_gain_weighted_terms_with_non_shipped_gains(tolfg, oracle, tmp_path, mission, dtype, pattern):
    """The shipped gains zero the S10 thrust terms (kT = 0) and G7's kp / kv terms; here kT, kp, kv, kdt are all
    non-zero and different, through a temporary root_path, so the objective value and every objective-gradient
    entry of both missions is live on the GPU (callback and batch, both patterns, both element types)."""
    import shutil
    import torch
    data = tolfg.lib().tolfg_default_root().decode()
    root = tmp_path / "root"
    shutil.copytree(data, root)
    gains = [0.37, 5.3, 2.9, 0.0, 1.7]
    (root / "problems" / mission / "gains.param").write_text("".join("%.17g // gain\n" % g for g in gains))
    N, B = 40, 5
    ops = [oracle.Problem(mission, "tempest", N=N, radius_goal=100.0 if mission == "S10" else 0.0, gains=gains, Vref=1.0 + t) for t in range(B)]
    X = np.stack([oracle.perturbed(ops[t], 300 + t) for t in range(B)])
    cidx = oracle.compact_index(ops[0])
    pick = (lambda G: G[cidx]) if pattern == "compact" else (lambda G: G)
    if dtype == "f64":           # the SNOPT callback computes in fp64 only
        p = tolfg.Problem(mission, "tempest", ts=N, radius_goal=100.0 if mission == "S10" else 0.0, Vref=1.0, root_path=str(root) + "/", pattern=pattern)
        F, G, st = p.define_fg(X[0])
        Fo, Go = ops[0].eval(X[0])
        assert st == 1 and Fo[0] != 0 and np.all(Go[:ops[0].c0] != 0)
        assert_close(F, Fo, what="gains callback F")
        assert_close(G, pick(Go), mask=pick(ops[0].undefined_mask()), what="gains callback G")
        p.close()
    bt = tolfg.Batch(mission, ["tempest"], ts=N, dtype=dtype, root_path=str(root) + "/", pattern=pattern)
    bt.set_trajectories([tolfg.Trajectory(radius_goal=100.0 if mission == "S10" else 0.0, Vref=1.0 + t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).to(bt.torch_dtype()).cuda()
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    Xin = dX[:, :bt.n].double().cpu().numpy()
    iG = bt.pattern()[0]
    for t in range(B):
        Fo, Go = ops[t].eval(Xin[t])
        Ft, Gt = dF[t, :bt.neF].double().cpu().numpy(), dG[t, :bt.neG].double().cpu().numpy()
        if dtype == "f64":
            assert_close(Ft, Fo, what=f"gains batch F[{t}]")
            assert_close(Gt, pick(Go), mask=pick(ops[t].undefined_mask()), what=f"gains batch G[{t}]")
        else:
            assert_close_f32(Ft, Gt, Fo, pick(Go), iG, N, mask=pick(ops[t].undefined_mask()), what=f"gains batch f32 [{t}]")


@pytest.mark.parametrize("mission", ["S10", "G7"])
def test_batched_need_flags_and_repeated_launches(tolfg, oracle, mission):
    """The tile-per-workgroup path (B > 8) honours needF / needG like the callback does, in any order of launches:
    the objective partial slots of the single-launch form are empty again after every launch (also after one
    that did not ask for F), and 30 further launches reproduce the first one bit for bit."""
    import torch
    N, B = 200, 20
    rg = 100.0 if mission == "S10" else 0.0
    bt = tolfg.Batch(mission, ["tempest", "skywalker"], ts=N)
    bt.set_trajectories([tolfg.Trajectory(aircraft=t % 2, radius_goal=rg, Vref=0.5 + 0.2 * t) for t in range(B)])
    ops = [oracle.Problem(mission, ("tempest", "skywalker")[t % 2], N=N, radius_goal=rg, Vref=0.5 + 0.2 * t) for t in range(B)]
    X = np.stack([oracle.perturbed(ops[t], 800 + t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).cuda()
    ref = [ops[t].eval(X[t]) for t in range(B)]

    def check(F_expected, G_expected):
        torch.cuda.synchronize()
        for t in range(B):
            if F_expected:
                assert_close(dF[t, :bt.neF].cpu().numpy(), ref[t][0], what=f"F[{t}]")
            else:
                assert torch.isnan(dF).all()
            if G_expected:
                assert_close(dG[t, :bt.neG].cpu().numpy(), ref[t][1], mask=ops[t].undefined_mask(), what=f"G[{t}]")
            else:
                assert torch.isnan(dG).all()

    for needF, needG in ((False, True), (True, False), (False, False), (True, True), (False, True), (True, True)):
        dF.fill_(float("nan")); dG.fill_(float("nan"))
        bt.eval(dX, dF, dG, needF=needF, needG=needG)
        check(needF, needG)
    F0, G0 = dF.clone(), dG.clone()
    for _ in range(30):
        bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    assert torch.equal(dF[:, :bt.neF], F0[:, :bt.neF]) and torch.equal(dG[:, :bt.neG], G0[:, :bt.neG])   # pad columns hold NaN


@pytest.mark.parametrize("mission,N", [("S10", 200), ("G7", 64), ("S10", 33)])
def test_callback_with_the_callers_own_arrays_kept_across_calls(tolfg, oracle, mission, N):
    """snOptA hands DEFINEGusrfg_ the SAME x, F and G arrays call after call.  With tolfg_config.persistent_arrays (the
    driver's promise that they stay put) the library registers an array it sees twice in a row and lets the kernel read
    x from it and write F and G into it (no staging copies); x changes in place between calls.  Every call must return
    the right numbers (ts = 33: n is odd, x then still goes through the pinned copy)."""
    import ctypes as C
    p = tolfg.Problem(mission, "skywalker", ts=N, radius_goal=100.0 if mission == "S10" else 0.0, persistent_arrays=True)
    o = oracle.Problem(mission, "skywalker", N=N, radius_goal=100.0 if mission == "S10" else 0.0)
    lib = tolfg.lib()
    x = np.ascontiguousarray(o.x0(), dtype=np.float64)
    F, G = np.full(p.neF, np.nan), np.full(p.neG, np.nan)
    dbl = C.POINTER(C.c_double)
    st, n, neF, neG = C.c_int(1), C.c_int(p.n), C.c_int(p.neF), C.c_int(p.neG)
    one, zero = C.c_int(1), C.c_int(0)
    rng = np.random.default_rng(3)
    p.make_current()
    for call in range(6):
        x += 0.003 * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))          # in place: same array, new values
        x[0] = abs(x[0]) + 0.01
        F[:] = np.nan; G[:] = np.nan
        lib.DEFINEGusrfg_(C.byref(st), C.byref(n), x.ctypes.data_as(dbl), C.byref(one), C.byref(neF), F.ctypes.data_as(dbl), C.byref(one),
                          C.byref(neG), G.ctypes.data_as(dbl), None, C.byref(zero), None, C.byref(zero), None, C.byref(zero))
        assert st.value == 1
        Fo, Go = o.eval(x)
        assert_close(F, Fo, what=f"call {call} F")
        assert_close(G, np.where(o.undefined_mask(), 0.0, Go), mask=o.undefined_mask(), what=f"call {call} G")
        assert p.registered_arrays() == (0 if call == 0 else (3 if p.n % 2 == 0 else 2))
    p.close()


def test_a_batch_moved_between_streams_keeps_its_evaluations_apart(tolfg, oracle):
    """Stream contract: the per-launch workspace belongs to one evaluation at a time.  A caller that alternates between
    two streams without synchronising gets the ordering enforced by the library (the previous stream is drained first):
    every evaluation's objective must be right -- a race on the partial slots or arrival counters would show there."""
    import torch
    N, B = 200, 64
    bt = tolfg.Batch("S10", ["tempest"], ts=N)
    bt.set_trajectories([tolfg.Trajectory(Vref=0.1 * t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    want = dF[:, :bt.neF].clone()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [bt.alloc(B)[1:] for _ in range(2)]
    for i in range(24):
        with torch.cuda.stream(streams[i & 1]):
            outs[i & 1][0].zero_()
            bt.eval(dX, outs[i & 1][0], outs[i & 1][1])
    torch.cuda.synchronize()
    bt.status()
    for F, _ in outs:
        assert torch.equal(F[:, :bt.neF], want)
    bt.close()
