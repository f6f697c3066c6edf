"""Worker of tests/test_multi_loopback.py: a FRESH process (the collective library is chosen once per process) that runs
the native several-devices host path with several parts on device 0 -- TOLFG_MULTI_SHARED_DEVICES=1 and the loop-back
collective library in TOLFG_RCCL_LIBRARY, both set by the parent -- and the same trajectories as ONE batch through the
single-GPU entry points; everything lands in an .npz for the parent to compare."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def device_list(parts):
    """Device ordinal of every part: all on device 0 under the shared-devices seam (loop-back collectives, one-GPU box), else
    one real device each (tests/test_multi_real_devices.py, a node with several GPUs)."""
    return [0] * parts if os.environ.get("TOLFG_MULTI_SHARED_DEVICES") == "1" else list(range(parts))


def trajectories(tolfg, mission, total):
    ms = [("S10", "G7")[t % 2] if mission == "mixed" else mission for t in range(total)]
    return [tolfg.Trajectory(aircraft=t % 2, mission=ms[t], radius_goal=100.0 if ms[t] == "S10" else 0.0, Vref=0.4 + 0.1 * t,
                             href=8.0 + 0.25 * t, xi=3.0 * t - 20.0, yi=-1.5 * t, zi=-35.0 - t) for t in range(total)]


class _Raw:
    """A device pointer as something torch can view (no ownership)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def pipeline(out, mission, dtype, total, parts, N, issue, gather="rccl"):
    """The asynchronous gather (tolfg_multi_step / gather_begin / gather_wait) against the synchronous one: `nsteps` steps on
    different inputs, the host waiting for gather j-1 only after it has issued step j, then the same inputs one by one through
    eval_from + gather_objectives.  Also: a ticket that has expired, the native step loop, the state after it."""
    import json
    import torch
    import tol_amd as tolfg
    air = ["tempest", "skywalker"]
    trajs = trajectories(tolfg, mission, total)
    devs = device_list(parts)
    m = tolfg.Multi(mission, air, ts=N, dtype=dtype, devices=devs)
    m.set_issue(issue)
    m.set_gather(gather)
    m.set_placement(2)
    m.set_trajectories(trajs)
    m.x0()
    m.sync()
    res = {"library": np.array(m.rccl_library()), "shards": np.array([m.shard(i) for i in range(parts)])}
    nsteps = 2 * tolfg.capi.MULTI_SLOTS + 3
    tdt = torch.float64 if dtype == "f64" else torch.float32
    sets, keep = [], []
    for j in range(nsteps):
        ptrs = []
        for i in range(parts):
            lo, hi = m.shard(i)
            (dX, ldx), _, _ = m.buffers(i)
            with torch.cuda.device(devs[i]):
                base = torch.as_tensor(_Raw(dX, (max(hi - lo, 1), ldx), "<f8" if dtype == "f64" else "<f4"), device=torch.device("cuda", devs[i]))
            assert base.data_ptr() == dX
            x = base.clone()
            x[:, 1:] *= 1.0 + 2e-3 * j           # set 0 = the initial guesses themselves
            keep.append(x)
            ptrs.append(x.data_ptr())
        sets.append(ptrs)
    for d in set(devs):
        torch.cuda.synchronize(d)
    tickets, got = [], []
    for j in range(nsteps):
        tickets.append(m.step(dX=sets[j]))
        if j >= 1:
            got.append(m.gather_wait(tickets[j - 1]))      # step j is issued: its evaluation runs beside this wait
    got.append(m.gather_wait(tickets[-1]))
    res["obj_async"] = np.stack(got)
    res["tickets"] = np.array(tickets)
    # a run of steps nobody waits for: the first ticket expires, the last is good
    run = [m.step(dX=sets[j % nsteps]) for j in range(tolfg.capi.MULTI_SLOTS + 2)]
    try:
        m.gather_wait(run[0])
        res["expired"] = np.array("no error")
    except tolfg.TolfgError as e:
        res["expired"] = np.array(f"{e.code} {e}")
    res["obj_run_last"] = m.gather_wait(run[-1])
    res["run_last_set"] = np.array((tolfg.capi.MULTI_SLOTS + 1) % nsteps)
    # the synchronous form on the same inputs
    sync = []
    for j in range(nsteps):
        m.eval_from(sets[j])
        sync.append(m.gather_objectives())
    res["obj_sync"] = np.stack(sync)
    res["mean_last"] = np.array(m.mean_objective())
    # a longer run over the rotating objective buffers: 400 steps, the host three tickets behind the step it issues (the slot-reuse
    # wait is exercised every step), every gathered vector compared with the synchronous one of its input set
    lag, bad, tk = tolfg.capi.MULTI_SLOTS - 1, 0, []
    for j in range(400):
        tk.append(m.step(dX=sets[j % nsteps]))
        if j >= lag:
            bad += not np.array_equal(m.gather_wait(tk[j - lag]), sync[(j - lag) % nsteps])
    for j in range(400 - lag, 400):
        bad += not np.array_equal(m.gather_wait(tk[j]), sync[j % nsteps])
    res["soak_mismatches"] = np.array(bad)
    # the native step loop, with and without the gather; then the object still works
    tim = [m.time_steps(12, warm=3, x_sets=sets[:5]), m.time_steps(12, warm=3, x_sets=sets[:5], gather=False), m.time_steps(5, warm=0)]
    res["timing"] = np.array(json.dumps(tim))
    m.eval_from(sets[1])
    res["obj_after_loop"] = m.gather_objectives()
    # One device's launch refuses to start (an earlier evaluation of its shard lost an objective partial: an x that carries the
    # empty-slot marker) while the other devices' launches and their calls of the collective go ahead: the step must come back
    # with the error -- no device left waiting for a rank that never joins -- and the next step must work.
    if mission == "mixed" and os.environ.get("TOLFG_FUSED") != "0":       # (the two-launch form, a measurement override, has no polled slots)
        marker = np.array([0xFFFBADADFFFBADAD], dtype=np.uint64).view(np.float64)[0]
        part = 2
        lo, hi = m.shard(part)
        (dXp, ldx), _, _ = m.buffers(part)
        poisoned = keep[part].clone()                                  # set 0's rows of that part
        poisoned[1, 1 + 11 * 7 + 10] = float(marker)                   # thrust of node 7 of its second trajectory
        assert poisoned[1, 1 + 11 * 7 + 10].view(torch.int64).item() == np.float64(marker).view(np.int64)
        bad_set = list(sets[0])
        bad_set[part] = poisoned.data_ptr()
        torch.cuda.synchronize(devs[part])
        m.step(dX=bad_set)                                             # runs; that shard's status word gets set
        m.sync()
        try:
            m.step(dX=sets[0])                                         # part 2 refuses (lost partial pending), the others launch and gather
            res["refusal"] = np.array("no error")
        except tolfg.TolfgError as e:
            res["refusal"] = np.array(f"{e.code} {e}")
        res["obj_after_refusal"] = m.gather_wait(m.step(dX=sets[0]))
    # the other way of gathering on the same object: the same numbers
    m.set_gather("host" if gather == "rccl" else "rccl")
    res["obj_other_gather"] = m.gather_wait(m.step(dX=sets[2]))
    res["mean_other_gather"] = np.array(m.mean_objective())
    m.close()
    # the single batch on set 0 (= the initial guesses)
    bt = tolfg.Batch(mission, air, ts=N, dtype=dtype)
    bt.set_trajectories(trajs)
    dX, dF, dG = bt.alloc(total)
    bt.x0_device(dX)
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    res["obj_single"] = dF[:, 0].cpu().numpy()
    bt.close()
    np.savez(out, **res)


def main():
    out, mission, dtype, total, parts, N, wind = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    if wind.startswith("pipeline:"):
        return pipeline(out, mission, dtype, total, parts, N, *wind.split(":")[1:])
    import torch
    import tol_amd as tolfg
    from helpers import random_wind_table
    air = ["tempest", "skywalker"]
    wm = tolfg.capi.WIND_TABLE if wind == "table" else tolfg.capi.WIND_SHEAR
    trajs = trajectories(tolfg, mission, total)
    m = tolfg.Multi(mission, air, ts=N, dtype=dtype, devices=device_list(parts), windmodel=wm)
    res = {"library": np.array(m.rccl_library())}
    m.set_trajectories(trajs)
    res["shards"] = np.array([m.shard(i) for i in range(parts)])
    tables = None
    if wind == "table":
        tables = np.stack([random_wind_table(N, 300 + t) for t in range(total)])
        m.set_wind_tables(tables)
        res["tables"] = tables
    m.x0()
    m.eval()
    res["obj"] = m.gather_objectives()
    res["mean"] = np.array(m.mean_objective())
    # a second evaluation + gather: the collectives are re-entrant and the partial slots were emptied
    m.eval()
    res["obj_again"] = m.gather_objectives()
    # the other way of gathering (the finalizing waves store into one pinned host vector), empty shards included
    m.set_gather("host")
    m.eval()
    res["obj_host"] = m.gather_objectives()
    res["mean_host"] = np.array(m.mean_objective())
    m.set_gather("rccl")
    for i in range(parts):
        lo, hi = m.shard(i)
        if hi > lo:
            res[f"X{i}"], res[f"F{i}"], res[f"G{i}"] = m.fetch(i, with_x=True)
    m.close()
    # the same trajectories as one batch
    bt = tolfg.Batch(mission, air, ts=N, dtype=dtype, windmodel=wm)
    bt.set_trajectories(trajs)
    dX, dF, dG = bt.alloc(total)
    bt.x0_device(dX)
    dW = None
    if tables is not None:
        dW = torch.from_numpy(tables if dtype == "f64" else tables.astype(np.float32)).cuda()
    bt.eval(dX, dF, dG, wind=dW)
    torch.cuda.synchronize()
    res["Xs"], res["Fs"], res["Gs"] = dX[:, :bt.n].cpu().numpy(), dF[:, :bt.neF].cpu().numpy(), dG[:, :bt.neG].cpu().numpy()
    res["sizes"] = np.array([bt.n, bt.neF, bt.neG])
    bt.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main()
