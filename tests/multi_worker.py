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


def trajectories(tolfg, mission, total):
    ms = [("S10", "G7")[t % 2] if mission == "mixed" else mission for t in range(total)]
    return [tolfg.Trajectory(aircraft=t % 2, mission=ms[t], radius_goal=100.0 if ms[t] == "S10" else 0.0, Vref=0.4 + 0.1 * t,
                             href=8.0 + 0.25 * t, xi=3.0 * t - 20.0, yi=-1.5 * t, zi=-35.0 - t) for t in range(total)]


def main():
    out, mission, dtype, total, parts, N, wind = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    import torch
    import tol_amd as tolfg
    from helpers import random_wind_table
    air = ["tempest", "skywalker"]
    wm = tolfg.capi.WIND_TABLE if wind == "table" else tolfg.capi.WIND_SHEAR
    trajs = trajectories(tolfg, mission, total)
    m = tolfg.Multi(mission, air, ts=N, dtype=dtype, devices=[0] * parts, windmodel=wm)
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
