"""The native several-GPUs-one-process host path (include/tolfg.h section 4; tol_amd/csrc/multi.cpp).

CPU: the sharding rule and the re-ordering of an all-gather's blocks equal tol_amd/distributed.py's (so the C++ and the
Python host paths split and order a batch the same way); creation without a GPU fails loudly; the C99 example compiles.
GPU (one device on the test box): the real RCCL calls -- ncclCommInitAll, grouped ncclAllGather, ncclAllReduce -- carry
the objectives of a sharded batch, whose numbers equal the oracle's.  More than one device: not measurable on this
pool's one-GPU boxes (the driver's 8-GPU run exercises the Python path, bench.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "examples", "batch_montecarlo_multi.c")


def build_example(tolfg, out):
    libdir = os.path.dirname(tolfg.lib_path())
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                    "-o", out, "-L", libdir, "-ltolfg", "-L", "/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)


@pytest.mark.parametrize("total,world", [(1024, 8), (8192, 8), (1001, 2), (7, 8), (13, 5), (1, 1), (0, 3)])
def test_shard_rule_and_gather_order_match_the_python_host_path(tolfg, total, world):
    from tol_amd.distributed import shard_bounds
    L = tolfg.lib()
    width = shard_bounds(total, 0, world)[1]
    padded = np.full(width * world, -1.0)
    covered = []
    for r in range(world):
        lo, hi = C.c_long(), C.c_long()
        assert L.tolfg_shard_bounds(total, r, world, C.byref(lo), C.byref(hi)) == 0
        assert (lo.value, hi.value) == shard_bounds(total, r, world)
        padded[r * width: r * width + hi.value - lo.value] = np.arange(lo.value, hi.value)
        covered += list(range(lo.value, hi.value))
    assert covered == list(range(total))
    out = np.full(max(total, 1), -2.0)
    assert L.tolfg_compact_gathered(padded.ctypes.data, 8, total, world, out.ctypes.data) == 0
    assert np.array_equal(out[:total], np.arange(total))
    p32 = padded.astype(np.float32)
    o32 = np.zeros(max(total, 1), dtype=np.float32)
    assert L.tolfg_compact_gathered(p32.ctypes.data, 4, total, world, o32.ctypes.data) == 0
    assert np.array_equal(o32[:total], np.arange(total, dtype=np.float32))
    lo, hi = C.c_long(), C.c_long()
    assert L.tolfg_shard_bounds(10, 3, 3, C.byref(lo), C.byref(hi)) == tolfg.capi.ERR_ARG


def test_example_is_plain_c(tolfg, tmp_path):
    build_example(tolfg, str(tmp_path / "mc_multi"))


def test_creation_without_a_gpu_fails_loudly(tolfg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for hosts without a GPU")
    m = tolfg.Multi.__new__(tolfg.Multi)
    with pytest.raises(tolfg.TolfgError) as e:
        tolfg.Multi.__init__(m, "S10", ["tempest"], ts=20, devices=[0])
    assert e.value.code == tolfg.capi.ERR_HIP


@pytest.mark.gpu
@pytest.mark.parametrize("mission,dtype,B", [("S10", "f64", 37), ("mixed", "f64", 24), ("G7", "f32", 16)])
def test_one_device_through_the_real_rccl_calls(tolfg, oracle, mission, dtype, B):
    import torch
    air = ["tempest", "skywalker"]
    ms = [("S10", "G7")[t % 2] if mission == "mixed" else mission for t in range(B)]
    trajs = [tolfg.Trajectory(aircraft=t % 2, mission=ms[t], radius_goal=100.0 if ms[t] == "S10" else 0.0, Vref=0.3 * t, href=9.0,
                              xi=2.0 * t, yi=-1.0 * t, zi=-40.0 - t) for t in range(B)]
    m = tolfg.Multi(mission, air, ts=100, dtype=dtype, devices=[0])
    assert "rccl" in m.rccl_library()
    m.set_trajectories(trajs)
    assert m.shard(0) == (0, B)
    m.x0()
    m.eval()
    obj = m.gather_objectives()
    mean = m.mean_objective()
    # the same batch through the single-GPU entry points
    bt = tolfg.Batch(mission, air, ts=100, dtype=dtype)
    bt.set_trajectories(trajs)
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    assert np.array_equal(obj, dF[:, 0].cpu().numpy())
    assert mean == pytest.approx(float(dF[:, 0].double().mean()), rel=1e-12)
    # F and G of the shard live in the library's device buffers: compare through a device-to-host copy
    Fm, Gm = m.fetch(0)
    assert np.array_equal(Fm[:, :bt.neF], dF[:, :bt.neF].cpu().numpy()) and np.array_equal(Gm[:, :bt.neG], dG[:, :bt.neG].cpu().numpy())
    # and against the oracle (fp64)
    if dtype == "f64":
        for t in (0, B - 1):
            o = oracle.Problem(ms[t], air[t % 2], N=100, radius_goal=trajs[t].radius_goal, Vref=trajs[t].Vref, href=9.0,
                               start=(trajs[t].xi, trajs[t].yi, trajs[t].zi))
            assert obj[t] == pytest.approx(o.eval(o.x0(), needG=False)[0][0], rel=1e-12)
    m.close()


@pytest.mark.gpu
def test_c_example_runs_on_one_device(tolfg, oracle, tmp_path):
    exe = str(tmp_path / "mc_multi")
    build_example(tolfg, exe)
    B = 41
    out = subprocess.run([exe, str(B), "0"], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pipelined 6 steps, objectives identical" in out.stdout
    w = out.stdout.split()
    first, last, mean = float(w[w.index("first") + 1]), float(w[w.index("last") + 1]), float(w[w.index("mean") + 1])
    vals = []
    for t in range(B):
        o = oracle.Problem("S10", "tempest", N=200, Vref=5.0 * t / (B - 1), href=10.0, start=(0.0, 0.0, -50.0))
        vals.append(o.eval(o.x0(), needG=False)[0][0])
    assert first == pytest.approx(vals[0], rel=1e-12) and last == pytest.approx(vals[-1], rel=1e-12)
    assert mean == pytest.approx(np.mean(vals), rel=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("wind", ["table", "grid"])
def test_wind_reaches_every_shard(tolfg, oracle, wind):
    """Per-trajectory wind tables are dealt to the shards in global order; a gridded field goes to every device."""
    import torch
    from helpers import random_wind_table
    from test_wind_grid import make_grid
    N, B = 64, 11
    trajs = [tolfg.Trajectory(aircraft=0, radius_goal=100.0, xi=37.0 + 2.0 * t, yi=-41.0 - t, zi=-45.0) for t in range(B)]
    m = tolfg.Multi("S10", ["tempest"], ts=N, devices=[0], windmodel=tolfg.capi.WIND_TABLE if wind == "table" else tolfg.capi.WIND_SHEAR)
    m.set_trajectories(trajs)
    bt = tolfg.Batch("S10", ["tempest"], ts=N, windmodel=tolfg.capi.WIND_TABLE if wind == "table" else tolfg.capi.WIND_SHEAR)
    bt.set_trajectories(trajs)
    dW = None
    if wind == "table":
        tables = np.stack([random_wind_table(N, 90 + t) for t in range(B)])
        m.set_wind_tables(tables)
        dW = torch.from_numpy(tables).cuda()
    else:
        g = make_grid(5)
        m.set_wind_grid(g["v"], g["origin"], g["spacing"], g["datum"])
        bt.set_wind_grid(g["v"], g["origin"], g["spacing"], g["datum"])
    m.x0()
    m.eval()
    obj = m.gather_objectives()
    Fm, Gm = m.fetch(0)
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    bt.eval(dX, dF, dG, wind=dW)
    torch.cuda.synchronize()
    assert np.array_equal(Fm[:, :bt.neF], dF[:, :bt.neF].cpu().numpy()) and np.array_equal(Gm[:, :bt.neG], dG[:, :bt.neG].cpu().numpy())
    assert np.array_equal(obj, dF[:, 0].cpu().numpy())
    # the wind really entered: the dt column of the x-defect rows carries -(W_x + Va e_x), different from the shear run
    ref = tolfg.Batch("S10", ["tempest"], ts=N)
    ref.set_trajectories(trajs)
    rF, rG = ref.alloc(B)[1:]
    ref.eval(dX, rF, rG)
    torch.cuda.synchronize()
    assert not torch.equal(rG[:, :ref.neG], dG[:, :bt.neG])
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("issue,gather", [("grouped", "rccl"), ("threads", "rccl"), ("grouped", "host"), ("threads", "host")])
def test_pipelined_steps_through_the_real_rccl_calls(tolfg, issue, gather):
    """tolfg_multi_step / gather_wait on one device with the real library: ncclAllGather on the gather stream (inside a group
    bracket, or as a plain call from the device's thread), or no collective at all (objectives stored to the pinned host vector);
    the native step loop; the same objectives as the synchronous gather every time."""
    B = 40
    trajs = [tolfg.Trajectory(aircraft=t % 2, radius_goal=100.0, Vref=0.3 * t, href=9.0, xi=2.0 * t, yi=-1.0 * t, zi=-40.0 - t) for t in range(B)]
    m = tolfg.Multi("S10", ["tempest", "skywalker"], ts=100, devices=[0])
    m.set_issue(issue)
    m.set_gather(gather)
    m.set_trajectories(trajs)
    m.x0()
    m.eval()
    want = m.gather_objectives()
    assert np.isfinite(want).all() and want[0] != want[1]
    tickets = [m.step() for _ in range(3)]                      # three steps in flight, nobody waiting
    got = [m.gather_wait(t) for t in tickets]
    for _ in range(10):                                         # beyond the four rotating buffers, one ticket behind
        tickets.append(m.step())
        got.append(m.gather_wait(tickets[-2]))
    got.append(m.gather_wait(tickets[-1]))
    assert all(np.array_equal(g, want) for g in got)
    t = m.time_steps(30, warm=5)
    assert t["issue"] == issue and t["gather"] == gather and t["wall_us_per_step"] > 0 and t["gather_us"] > 0
    assert m.mean_objective() == pytest.approx(float(want.mean()), rel=1e-12)
    with pytest.raises(tolfg.TolfgError):
        m.gather_wait(tickets[0])                               # long expired
    m.close()
