"""BASELINE.json configs[1..4] at their full sizes on the GPU, through the C ABI: oracle comparison
where the oracle finishes in seconds, size-independent exact properties everywhere."""
import os

import numpy as np
import pytest

from helpers import assert_close_f32_batch, assert_close, assert_close_f32

pytestmark = pytest.mark.gpu

AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]

# structure of one 104-entry slab, 8 rows x [dt | x y z Va gam chi phi CL dphi dCL T | next]
_ONE = [13 * r + 12 for r in range(8)]
_MINUS_ONE = [1, 15, 29, 85, 99]
_COMPUTED = [0, 4, 5, 6, 13, 17, 18, 19, 26, 30, 31, 39, 43, 44, 45, 47, 50, 52, 56, 57, 58, 59, 60,
             65, 69, 70, 71, 72, 73, 78, 87, 91, 101]
_ZERO = sorted(set(range(104)) - set(_ONE) - set(_MINUS_ONE) - set(_COMPUTED))


def check_exact_structure(G, c0, N, x):
    """58 structural zeros, 8 ones, 5 minus-ones per slab, and -dt in the two control columns: exact."""
    slabs = G[:, c0:c0 + 104 * N].reshape(G.shape[0], N, 104)
    assert len(_ZERO) == 58
    assert (slabs[:, :, _ZERO] == 0.0).all()
    assert (slabs[:, :, _ONE] == 1.0).all()
    assert (slabs[:, :, _MINUS_ONE] == -1.0).all()
    assert (slabs[:, :, 87] == -x[:, :1]).all() and (slabs[:, :, 101] == -x[:, :1]).all()
    node = x[:, 1:].reshape(x.shape[0], N + 1, 11)
    assert (slabs[:, :, 78] == -node[:, :N, 8]).all()      # d defect_phi / d dt = -phidot
    assert (slabs[:, :, 91] == -node[:, :N, 9]).all()      # d defect_CL  / d dt = -CLdot


def test_config2_single_trajectory_s10_tempest_200(tolfg, oracle):
    p = tolfg.Problem("S10", "tempest", ts=200)
    o = oracle.Problem("S10", "tempest", N=200)
    for seed in (7, 8, 9):
        x = oracle.perturbed(o, seed)
        F, G, st = p.define_fg(x)
        Fo, Go = o.eval(x)
        assert st == 1
        assert_close(F, Fo, what="cfg2 F")
        assert_close(G, Go, mask=o.undefined_mask(), what="cfg2 G")
        check_exact_structure(G[None, :], o.c0, 200, x[None, :])
    p.close()


def test_config3_single_trajectory_s10_skywalker_2000(tolfg, oracle):
    N = 2000
    p = tolfg.Problem("S10", "skywalker", ts=N)
    o = oracle.Problem("S10", "skywalker", N=N)
    assert (p.n, p.neF, p.neG) == (22012, 16012, 214037)          # SURVEY.md section 8, probed
    for x in (o.x0(), oracle.perturbed(o, 7)):
        F, G, st = p.define_fg(x)
        Fo, Go = o.eval(x)
        assert st == 1
        assert_close(F, Fo, what="cfg3 F")
        assert_close(G, Go, mask=o.undefined_mask(), what="cfg3 G")
        check_exact_structure(G[None, :], o.c0, N, x[None, :])
    # defects are affine in the next node's state with unit slope: moving x_{k+1,r} by d moves defect r of node k by d
    x = oracle.perturbed(o, 3)
    F0 = p.define_fg(x, needG=False)[0]
    x2 = x.copy()
    k, r, d = 1234, 4, 0.5
    x2[11 * (k + 1) + 1 + r] += d
    F1 = p.define_fg(x2, needG=False)[0]
    assert F1[8 * k + 1 + r] - F0[8 * k + 1 + r] == pytest.approx(d, abs=1e-12)
    p.close()


def _batch_inputs(tolfg, oracle, mission, B, N, seed, n_aircraft=1):
    rng = np.random.default_rng(seed)
    trajs = []
    for t in range(B):
        trajs.append(tolfg.Trajectory(aircraft=t % n_aircraft, Vref=rng.uniform(0, 5), href=rng.uniform(5, 20),
                                      radius_goal=100.0 if mission == "S10" else 0.0,
                                      xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50)))
    return trajs, rng.uniform(-100, -20, B)


def _oracle_for(oracle, mission, tr, zi, N):
    return oracle.Problem(mission, AIRCRAFT[tr.aircraft], N=N, east_goal=tr.east_goal, north_goal=tr.north_goal,
                          radius_goal=tr.radius_goal, start=(tr.xi, tr.yi, zi), Vref=tr.Vref, href=tr.href)


def test_config4_batch_1024_s10_randomised_wind_and_start(tolfg, oracle):
    import torch
    B, N = 1024, 200
    bt = tolfg.Batch("S10", ["tempest"], ts=N)
    trajs, zis = _batch_inputs(tolfg, oracle, "S10", B, N, 1000)
    bt.set_trajectories(trajs)
    X = np.empty((B, bt.n))
    for t in range(B):
        rng = np.random.default_rng(5000 + t)
        x = bt.x0(t, zi=zis[t])
        X[t] = x + 0.05 * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))
        X[t, 0] = abs(X[t, 0]) + 0.01
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).cuda()
    obj = torch.empty(B, dtype=torch.float64, device="cuda")
    bt.eval(dX, dF, dG, obj=obj)
    torch.cuda.synchronize()
    F, G = dF[:, :bt.neF].cpu().numpy(), dG[:, :bt.neG].cpu().numpy()
    assert np.array_equal(obj.cpu().numpy(), F[:, 0])
    worst = 0.0
    for t in range(B):                       # the whole batch against the oracle
        o = _oracle_for(oracle, "S10", trajs[t], zis[t], N)
        Fo, Go = o.eval(X[t])
        worst = max(worst, assert_close(F[t], Fo, what=f"cfg4 F[{t}]"),
                    assert_close(G[t], Go, mask=o.undefined_mask(), what=f"cfg4 G[{t}]"))
    check_exact_structure(G, 3 * N + 4, N, X)
    # periodic boundary rows are plain differences: exact
    want = X[:, 11 * N + 1:11 * N + 12] - X[:, 1:12]
    want[:, 5] -= 2 * np.pi
    assert np.array_equal(F[:, 8 * N + 1:], want)


def _mixed_inputs(tolfg, B, seed):
    """BASELINE configs[4]: mission = b mod 2 (S10, G7), air-frame = b mod 5, randomised shear wind and start."""
    rng = np.random.default_rng(seed)
    trajs = []
    for t in range(B):
        mission = ("S10", "G7")[t % 2]
        trajs.append(tolfg.Trajectory(aircraft=t % 5, mission=mission, Vref=rng.uniform(0, 5), href=rng.uniform(5, 20),
                                      radius_goal=100.0 if mission == "S10" else 0.0,
                                      xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50)))
    return trajs, rng.uniform(-100, -20, B)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_config5_mixed_missions_all_airframes_8192(tolfg, oracle, dtype):
    """ONE batch object, ONE launch: 8192 trajectories, S10 and G7 interleaved, all five air-frames."""
    import torch
    N, B = 200, 8192
    bt = tolfg.Batch("mixed", AIRCRAFT, ts=N, dtype=dtype)
    sizes = {m: bt.sizes_of(m) for m in ("S10", "G7")}
    assert (bt.n, bt.neF, bt.neG) == (sizes["S10"][0], sizes["G7"][1], sizes["S10"][2])      # rows sized for the larger
    trajs, zis = _mixed_inputs(tolfg, B, 77)
    bt.set_trajectories(trajs)
    X = np.empty((B, bt.n))
    for t in range(B):
        rng = np.random.default_rng(9000 + t)
        x = bt.x0(t, zi=zis[t])
        X[t] = x + 0.05 * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))
        X[t, 0] = abs(X[t, 0]) + 0.01
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).to(bt.torch_dtype()).cuda()
    dF.fill_(float("nan")); dG.fill_(float("nan"))
    obj = torch.empty(B, dtype=bt.torch_dtype(), device="cuda")
    bt.eval(dX, dF, dG, obj=obj)
    torch.cuda.synchronize()
    Xs = dX[:, :bt.n].double().cpu().numpy()
    Fa, Ga = dF.double().cpu().numpy(), dG.double().cpu().numpy()
    assert np.array_equal(obj.double().cpu().numpy(), Fa[:, 0])
    pats = {m: bt.pattern(m)[0] for m in ("S10", "G7")}
    worst = {}
    for m, off in (("S10", 0), ("G7", 1)):
        n, neF, neG = sizes[m]
        F, G = Fa[off::2, :neF], Ga[off::2, :neG]
        assert np.isfinite(F).all() and np.isfinite(G).all()
        # a row's tail beyond its own mission's sizes is never written
        assert np.isnan(Fa[off::2, neF:]).all() and np.isnan(Ga[off::2, neG:]).all()
        check_exact_structure(G, 3 * N + 4 if m == "S10" else N + 6, N, Xs[off::2])
        # ALL 4096 trajectories of the mission against the oracle, evaluated as a batch on the host's cores (fp64 since round 4; the
        # fp32 leg since round 5 -- until then every 64th trajectory was compared and the rest only met the structure checks above).
        # fp32: the oracle is evaluated at the float32-rounded inputs the kernel saw, and compared per row class.
        idx = list(range(off, B, 2))
        probs = [_oracle_for(oracle, m, trajs[t], zis[t], N) for t in idx]
        Fo, Go, _ = oracle.eval_batch(probs, Xs[idx], nthreads=min(16, len(os.sched_getaffinity(0))))
        mask = probs[0].undefined_mask()
        if dtype == "f64":
            assert_close(Fa[idx, :neF], Fo, what=f"cfg5 {m} F (all {len(idx)} trajectories)")
            assert_close(Ga[idx, :neG], Go, mask=np.broadcast_to(mask, Go.shape), what=f"cfg5 {m} G (all {len(idx)} trajectories)")
        else:
            w = assert_close_f32_batch(Fa[idx, :neF], Ga[idx, :neG], Fo, Go, pats[m], N, mask=mask, what=f"cfg5 {m} f32 (all {len(idx)} trajectories)")
            for k, v in w.items():
                worst[(m,) + k] = max(worst.get((m,) + k, 0.0), v)
    if worst:
        print("fp32 worst scaled error per class:", {k: f"{v:.2e}" for k, v in sorted(worst.items())})


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_mixed_batch_equals_single_mission_batches(tolfg, oracle, dtype):
    """A mixed batch gives every row the numbers its single-mission batch gives (the mission branch is
    wave-uniform), including the device-side initial guess and bounds.  Not bitwise: the mixed kernels are
    separate template instances, so the compiler may contract different multiply-adds (last-bit differences)."""
    import torch
    N, B = 52, 40
    tight = 1e-13 if dtype == "f64" else 2e-6
    trajs, zis = _mixed_inputs(tolfg, B, 5)
    for t, tr in enumerate(trajs):
        tr.zi = float(zis[t])
    mixed = tolfg.Batch("mixed", AIRCRAFT, ts=N, dtype=dtype)
    mixed.set_trajectories(trajs)
    dX, dF, dG = mixed.alloc(B)
    mixed.x0_device(dX)
    xl, xu, Fl, Fu = (torch.full_like(dX, float("nan")), torch.full_like(dX, float("nan")),
                      torch.full_like(dF, float("nan")), torch.full_like(dF, float("nan")))
    mixed.bounds_device(xl, xu, Fl, Fu)
    x0_mixed = dX.clone()
    dX[:, 1:mixed.n] *= 1.003
    mixed.eval(dX, dF, dG)
    torch.cuda.synchronize()
    for m, off in (("S10", 0), ("G7", 1)):
        one = tolfg.Batch(m, AIRCRAFT, ts=N, dtype=dtype)
        sub = trajs[off::2]
        one.set_trajectories(sub)
        sX, sF, sG = one.alloc(len(sub))
        one.x0_device(sX)
        sxl, sxu, sFl, sFu = torch.empty_like(sX), torch.empty_like(sX), torch.empty_like(sF), torch.empty_like(sF)
        one.bounds_device(sxl, sxu, sFl, sFu)
        flat = lambda t: t.double().cpu().numpy().ravel()      # noqa: E731
        assert_close(flat(x0_mixed[off::2, :one.n]), flat(sX[:, :one.n]), tol=tight, what=f"x0 {m}")
        sX[:, :one.n] = dX[off::2, :one.n]            # the same decision vectors as the mixed batch saw
        one.eval(sX, sF, sG)
        torch.cuda.synchronize()
        assert_close(flat(dF[off::2, :one.neF]), flat(sF[:, :one.neF]), tol=tight, what=f"F {m}")
        assert_close(flat(dG[off::2, :one.neG]), flat(sG[:, :one.neG]), tol=tight, what=f"G {m}")
        # bounds are constants: bitwise
        assert torch.equal(xl[off::2, :one.n], sxl[:, :one.n]) and torch.equal(xu[off::2, :one.n], sxu[:, :one.n])
        assert torch.equal(Fl[off::2, :one.neF], sFl[:, :one.neF]) and torch.equal(Fu[off::2, :one.neF], sFu[:, :one.neF])
        # host-side per-trajectory set-up of the mixed batch uses the trajectory's own mission
        t = off + 2
        assert np.array_equal(mixed.x0(t, zi=zis[t])[:one.n], one.x0(1, zi=zis[t]))
        for a, b in zip(mixed.bounds(t, zi=zis[t]), one.bounds(1, zi=zis[t])):
            assert np.array_equal(a[:len(b)], b)


@pytest.mark.parametrize("B", [1, 3, 8, 9])
@pytest.mark.parametrize("N", [1, 4, 5, 64, 68])
def test_ragged_batches_and_tile_edges(tolfg, oracle, B, N):
    import torch
    bt = tolfg.Batch("G7", ["tempest", "skywalker"], ts=N)
    trajs = [tolfg.Trajectory(aircraft=t % 2, radius_goal=0.0, Vref=1.0 + t, href=9.0) for t in range(B)]
    bt.set_trajectories(trajs)
    ops = [oracle.Problem("G7", ("tempest", "skywalker")[t % 2], N=N, radius_goal=0.0, Vref=1.0 + t, href=9.0) for t in range(B)]
    X = np.stack([oracle.perturbed(ops[t], 40 + t) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).cuda()
    dF.fill_(float("nan")); dG.fill_(float("nan"))
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    for t in range(B):
        Fo, Go = ops[t].eval(X[t])
        assert_close(dF[t, :bt.neF].cpu().numpy(), Fo, what=f"ragged F[{t}]")
        assert_close(dG[t, :bt.neG].cpu().numpy(), Go, what=f"ragged G[{t}]")


def test_bitwise_reproducible(tolfg, oracle):
    """No atomics, fixed summation order: two evaluations of the same inputs are bitwise identical."""
    import torch
    N, B = 200, 300
    bt = tolfg.Batch("S10", AIRCRAFT, ts=N)
    trajs, zis = _batch_inputs(tolfg, oracle, "S10", B, N, 5, n_aircraft=5)
    bt.set_trajectories(trajs)
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    dX[:, 1:bt.n] += 0.01 * torch.randn(B, bt.n - 1, dtype=torch.float64, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    outs = []
    for _ in range(3):
        dF.zero_(); dG.zero_()
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        outs.append((dF.clone(), dG.clone()))
    for F, G in outs[1:]:
        assert torch.equal(F, outs[0][0]) and torch.equal(G, outs[0][1])


@pytest.mark.parametrize("mission,B,N", [("S10", 2, 10000), ("G7", 20000, 4), ("S10", 1, 20001)])
def test_extreme_shapes(tolfg, oracle, mission, B, N):
    """Very long single trajectories and very many very short ones (odd ts falls back to scalar stores)."""
    import torch
    rg = 100.0 if mission == "S10" else 0.0
    bt = tolfg.Batch(mission, ["tempest"], ts=N)
    bt.set_trajectories([tolfg.Trajectory(radius_goal=rg, Vref=1.0 + (t % 7)) for t in range(B)])
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    dX[:, 1:bt.n] *= 1.001
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    for t in sorted({0, B // 2, B - 1}):
        o = oracle.Problem(mission, "tempest", N=N, radius_goal=rg, Vref=1.0 + (t % 7))
        x = dX[t, :bt.n].cpu().numpy()
        Fo, Go = o.eval(x)
        assert_close(dF[t, :bt.neF].cpu().numpy(), Fo, what=f"extreme F[{t}]")
        assert_close(dG[t, :bt.neG].cpu().numpy(), Go, mask=o.undefined_mask(), what=f"extreme G[{t}]")


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mission,N", [("S10", 5), ("G7", 6), ("S10", 7), ("G7", 200), ("S10", 201)])
def test_slab_stream_at_every_alignment(tolfg, oracle, mission, N, dtype):
    """The Jacobian slabs are streamed with 16-byte stores wherever the slab regions sit relative to a 16-byte
    boundary: odd c0 (odd ts), c0 % 4 == 2 in fp32 (G7 at ts = 200), and a G whose rows start 1, 2 or 3 elements past
    a boundary.  Elements before and after every row's G are guard values and must survive."""
    import torch
    B = 11
    rg = 100.0 if mission == "S10" else 0.0
    bt = tolfg.Batch(mission, ["tempest"], ts=N, dtype=dtype)
    bt.set_trajectories([tolfg.Trajectory(radius_goal=rg, Vref=1.0 + 0.3 * t) for t in range(B)])
    ops = [oracle.Problem(mission, "tempest", N=N, radius_goal=rg, Vref=1.0 + 0.3 * t) for t in range(B)]
    X = np.stack([oracle.perturbed(ops[t], 60 + t) for t in range(B)])
    dX, dF, _ = bt.alloc(B)
    dX[:, :bt.n] = torch.from_numpy(X).to(bt.torch_dtype()).cuda()
    Xin = dX[:, :bt.n].double().cpu().numpy()
    ref = [ops[t].eval(Xin[t]) for t in range(B)]
    iG = bt.pattern()[0]
    vmax = 2 if dtype == "f64" else 4
    ld = (bt.neG + 2 * vmax + vmax - 1) // vmax * vmax
    for k in range(vmax):
        big = torch.full((B, ld), -777.0, dtype=bt.torch_dtype(), device="cuda")
        dG = big[:, k:k + bt.neG + 1]            # rows start k elements past a 16-byte boundary; one guard column inside the view
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        assert torch.all(big[:, :k] == -777.0) and torch.all(big[:, k + bt.neG:] == -777.0), f"guard elements overwritten (offset {k})"
        for t in range(B):
            Ft, Gt = dF[t, :bt.neF].double().cpu().numpy(), big[t, k:k + bt.neG].double().cpu().numpy()
            if dtype == "f64":
                assert_close(Ft, ref[t][0], what=f"F[{t}] offset {k}")
                assert_close(Gt, ref[t][1], mask=ops[t].undefined_mask(), what=f"G[{t}] offset {k}")
            else:
                assert_close_f32(Ft, Gt, ref[t][0], ref[t][1], iG, N, mask=ops[t].undefined_mask(), what=f"f32 [{t}] offset {k}")


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("mission,tail", [("S10", "7:16"), ("mixed", "20:32"), ("G7", "33:8")])
def test_finer_tiled_tail_gives_the_same_results(tolfg, measure, oracle, monkeypatch, mission, tail, fused):
    """TOLFG_TAIL=count:nt (a measurement knob, off by default: profiles/r02_tail_tiles.md) cuts the last `count`
    trajectories of a launch into finer tiles.  Defects and the Jacobian must be bitwise what the plain tiling
    gives; the objective is summed over different tile partials, so 1e-13."""
    import torch
    N, B = 200, 33
    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        bt = tolfg.Batch(mission, AIRCRAFT, ts=N, library=measure)
        ms = [("S10", "G7")[t % 2] if mission == "mixed" else mission for t in range(B)]
        bt.set_trajectories([tolfg.Trajectory(aircraft=t % 5, mission=ms[t], radius_goal=100.0 if ms[t] == "S10" else 0.0,
                                              Vref=1.0 + 0.1 * t) for t in range(B)])
        dX, dF, dG = bt.alloc(B)
        bt.x0_device(dX)
        dX[:, 1:bt.n] += 0.01 * torch.randn(B, bt.n - 1, dtype=torch.float64, device="cuda", generator=torch.Generator("cuda").manual_seed(11))
        dF.zero_(); dG.zero_()
        for _ in range(2):                                   # twice: the arrival counters must be left at zero
            bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        out = dF.cpu().numpy().copy(), dG.cpu().numpy().copy()
        bt.close()
        for k in env:
            monkeypatch.delenv(k)
        return out
    F0, G0 = run({"TOLFG_FUSED": fused})
    F1, G1 = run({"TOLFG_FUSED": fused, "TOLFG_TAIL": tail})
    assert np.array_equal(G0, G1, equal_nan=True)
    assert np.array_equal(F0[:, 1:], F1[:, 1:], equal_nan=True)
    assert_close(F1[:, 0], F0[:, 0], tol=1e-13, what="objective with a finer-tiled tail")
