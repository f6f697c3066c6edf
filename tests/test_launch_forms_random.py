"""The single-launch form (the last-arriving tile wave finalizes: arrival counters, polled partial slots) against the
two-launch form (finalize_kernel) on random shapes, twice per shape on the same objects -- the concurrency-sensitive part
of the batched path.  Defects, boundary rows and the Jacobian must be bitwise equal; the objective sums the same partials
in the same order in both forms, so it must be bitwise equal too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]


def same(a, b):
    """Bitwise equality where NaN equals NaN (the pad elements of a mixed batch's rows keep the fill value)."""
    return bool(((a == b) | (torch_isnan(a) & torch_isnan(b))).all())


def torch_isnan(t):
    return t != t


@pytest.mark.parametrize("seed", range(6))
def test_single_launch_equals_two_launch_on_random_shapes(tolfg, measure, monkeypatch, seed):
    import torch
    rng = np.random.default_rng(900 + seed)
    for case in range(6):
        mission = ("S10", "G7", "mixed")[int(rng.integers(3))]
        N = int(rng.choice([1, 3, 17, 52, 53, 64, 65, 100, 128, 200, 257]))
        B = int(rng.choice([1, 2, 9, 10, 33, 127, 600, 1500]))
        dtype = ("f64", "f32")[int(rng.integers(2))]
        pattern = ("reference", "compact")[int(rng.integers(2))]
        trajs = [tolfg.Trajectory(aircraft=t % 5, mission=("S10", "G7")[t % 2] if mission == "mixed" else mission,
                                  radius_goal=100.0 if (mission == "S10" or (mission == "mixed" and t % 2 == 0)) else 0.0,
                                  Vref=1.0 + 0.01 * (t % 97), xi=float(t % 7), zi=-30.0 - (t % 11)) for t in range(B)]
        outs = []
        for fused in ("1", "0"):
            monkeypatch.setenv("TOLFG_FUSED", fused)
            monkeypatch.setenv("TOLFG_NO_SINGLE_LAUNCH", "1")        # small batches through the tile-per-workgroup kernels too
            bt = tolfg.Batch(mission, AIRCRAFT, ts=N, dtype=dtype, pattern=pattern, library=measure)
            bt.set_trajectories(trajs)
            dX, dF, dG = bt.alloc(B)
            bt.x0_device(dX)
            gen = torch.Generator(device="cuda").manual_seed(int(1000 * seed + case))
            dX[:, 1:bt.n] += (0.01 * torch.randn(B, bt.n - 1, dtype=torch.float64, device="cuda", generator=gen)).to(dX.dtype)
            res = []
            for rep in range(2):                                     # the second evaluation finds the counters and slots as the first left them
                dF.fill_(float("nan")); dG.fill_(float("nan"))
                bt.eval(dX, dF, dG)
                torch.cuda.synchronize()
                res.append((dF[:, :bt.neF].clone(), dG[:, :bt.neG].clone()))
            assert same(res[0][0], res[1][0]) and same(res[0][1], res[1][1]), (mission, N, B, dtype, pattern, fused)
            assert torch.isfinite(res[0][0][:, :bt.neF - 1]).all()    # (a mixed batch pads its S10 rows by one element)
            outs.append(res[0])
            bt.close()
        what = (mission, N, B, dtype, pattern)
        assert same(outs[0][1], outs[1][1]), what
        assert same(outs[0][0], outs[1][0]), what


@pytest.mark.parametrize("seed", range(4))
def test_rows_through_lds_in_two_passes_equal_one_pass(tolfg, measure, monkeypatch, seed):
    """FgArgs::sub_nodes = 32 (a tile's Jacobian rows go through LDS 32 nodes at a time, TOLFG_SUB_NODES) against the
    whole-tile form: bitwise the same F and G, for slab regions on and off 16-byte boundaries (odd ts; G7 rows in fp32), both
    patterns, tiles of 33..64 nodes, short last tiles, and a G buffer that itself sits 0..15 elements off a 64-byte boundary."""
    import torch
    rng = np.random.default_rng(4400 + seed)
    for case in range(7):
        mission = ("S10", "G7", "mixed")[int(rng.integers(3))]
        N = int(rng.choice([33, 34, 52, 53, 63, 64, 65, 97, 100, 128, 199, 200, 257]))
        B = int(rng.choice([1, 3, 10, 33, 127, 700]))
        dtype = ("f64", "f32")[int(rng.integers(2))]
        pattern = ("reference", "compact")[int(rng.integers(2))]
        off = int(rng.integers(0, 16))
        trajs = [tolfg.Trajectory(aircraft=t % 5, mission=("S10", "G7")[t % 2] if mission == "mixed" else mission,
                                  radius_goal=100.0 if (mission == "S10" or (mission == "mixed" and t % 2 == 0)) else 0.0,
                                  Vref=1.0 + 0.01 * (t % 97), xi=float(t % 7), zi=-30.0 - (t % 11)) for t in range(B)]
        outs = []
        for sub in ("0", "32"):
            monkeypatch.setenv("TOLFG_SUB_NODES", sub)
            monkeypatch.setenv("TOLFG_NO_SINGLE_LAUNCH", "1")
            bt = tolfg.Batch(mission, AIRCRAFT, ts=N, dtype=dtype, pattern=pattern, library=measure)
            bt.set_trajectories(trajs)
            dX, dF, dGfull = bt.alloc(B)
            ld = dGfull.shape[1]
            flat = torch.full((B * ld + 24,), float("nan"), dtype=dGfull.dtype, device="cuda")
            dG = flat[off:off + B * ld].view(B, ld)                  # G off a 16-byte boundary by `off` elements
            bt.x0_device(dX)
            gen = torch.Generator(device="cuda").manual_seed(int(77 * seed + case))
            dX[:, 1:bt.n] += (0.01 * torch.randn(B, bt.n - 1, dtype=torch.float64, device="cuda", generator=gen)).to(dX.dtype)
            dF.fill_(float("nan"))
            bt.eval(dX, dF, dG)
            torch.cuda.synchronize()
            outs.append((dF[:, :bt.neF].clone(), dG[:, :bt.neG].clone(), flat[:off].clone(), flat[off + B * ld:].clone()))
            bt.close()
        what = (mission, N, B, dtype, pattern, off)
        assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1]), what
        assert torch_isnan(outs[1][2]).all() and torch_isnan(outs[1][3]).all(), what     # nothing written outside G


def test_first_evaluation_on_a_non_blocking_stream_finalizes_every_trajectory(tolfg):
    """The first evaluation of a batch object sets up its workspace (arrival counters, empty partial slots).  That set-up
    is ordered on the LAUNCH stream: on the null stream it was not ordered against a non-blocking stream, and about one
    first evaluation in 25 left a trajectory without its finalizing wave (objective and boundary rows unwritten) -- found by
    tests/test_multi_loopback.py, whose parts launch on such streams.  Many fresh batch objects, each evaluated once."""
    import torch
    trajs = [tolfg.Trajectory(aircraft=t % 2, mission=("S10", "G7")[t % 2], radius_goal=100.0 * (1 - t % 2), Vref=1.0 + t) for t in range(6)]
    stream = torch.cuda.Stream()          # non-blocking with respect to the null stream
    src = tolfg.Batch("mixed", ["tempest", "skywalker"], ts=200)
    src.set_trajectories(trajs)
    dX, dF, dG = src.alloc(6)
    src.x0_device(dX)
    torch.cuda.synchronize()
    ref = None
    for trial in range(80):
        bt = tolfg.Batch("mixed", ["tempest", "skywalker"], ts=200)
        bt.set_trajectories(trajs)
        obj = torch.full((6,), float("nan"), dtype=torch.float64, device="cuda")
        dF.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            bt.eval(dX, dF, dG, obj=obj)
        stream.synchronize()
        got = (obj.clone(), dF[:, 0].clone(), dF[:, 1601:1612].clone())
        assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all() and torch.isfinite(got[2]).all(), (trial, got[0])
        if ref is None:
            ref = got
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), trial
        bt.status()
        bt.close()
    src.close()


@pytest.mark.parametrize("wind", ["table", "grid"])
def test_set_up_followed_at_once_by_an_evaluation_on_a_non_blocking_stream(tolfg, wind):
    """VERDICT r4 item 5, the audit of the race above: everything an evaluation reads besides X -- the trajectory table, a wind
    table, a wind grid -- is uploaded either on the launch stream (tolfg_batch_eval's first call) or by a call that drains the
    device before it returns (set_wind_grid, set_wind_table), never by null-stream work a non-blocking launch could overtake.
    Fresh objects, described and evaluated at once on a non-blocking stream, no synchronisation in between; every one must give
    the first one's numbers."""
    import torch
    from helpers import random_wind_table
    from test_wind_grid import make_grid
    N, B = 64, 6
    stream = torch.cuda.Stream()
    g = make_grid(5)
    tables = np.stack([random_wind_table(N, 40 + t) for t in range(B)])
    ref = None
    for trial in range(80):
        # the trajectories differ from object to object, so that a stale table of the object before would show
        k = trial % 2
        trajs = [tolfg.Trajectory(aircraft=0, radius_goal=100.0, xi=37.0 + 2.0 * t + k, yi=-41.0 - t, zi=-45.0, Vref=1.0 + t + k) for t in range(B)]
        bt = tolfg.Batch("S10", ["tempest"], ts=N, windmodel=tolfg.capi.WIND_TABLE if wind == "table" else tolfg.capi.WIND_SHEAR)
        bt.set_trajectories(trajs)
        dW = None
        if wind == "grid":
            bt.set_wind_grid(g["v"] * (1.0 + k), g["origin"], g["spacing"], g["datum"])
        with torch.cuda.stream(stream):
            X, F, G = bt.alloc(B)
            if wind == "table":
                dW = torch.from_numpy(tables * (1.0 + k)).to("cuda", non_blocking=True)
            bt.x0_device(X)
            bt.eval(X, F, G, wind=dW)
        stream.synchronize()
        got = (F[:, :bt.neF].clone(), G[:, :bt.neG].clone())
        assert torch.isfinite(got[0]).all()
        if trial < 2:
            ref = (ref or []) + [got]
        else:
            assert torch.equal(got[0], ref[k][0]) and torch.equal(got[1], ref[k][1]), trial
        bt.status()
        bt.close()
    assert not torch.equal(ref[0][0], ref[1][0])
