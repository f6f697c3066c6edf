#!/usr/bin/env python3
"""Time of tolfg_batch_x0_device (initial guesses generated on the device) for the bench's batches: the node-parallel
kernel against the serial reference form (TOLFG_X0_SERIAL=1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("TOLFG_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tol_amd", "lib", "libtolfg_measure.so"))   # the TOLFG_* switches exist in the measurement build only (tol_amd/csrc/knobs.h)
import tol_amd
import bench as BN

for mission, dtype, B, ts in (("mixed", "f64", 8192, 200), ("S10", "f64", 8192, 200), ("G7", "f64", 8192, 200), ("mixed", "f32", 8192, 200),
                              ("S10", "f64", 1024, 200), ("S10", "f64", 64, 2000), ("S10", "f64", 1, 200)):
    air = BN.AIRCRAFT5 if mission == "mixed" else ("tempest",)
    bt = tol_amd.Batch(mission, air, ts=ts, dtype=dtype)
    bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, mission, len(air)))
    dX, _, _ = bt.alloc(B)
    out = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bt.x0_device(dX)                  # the first call also uploads the node times and fills the per-(mission, node) table
    torch.cuda.synchronize()
    first = 1e6 * (time.perf_counter() - t0)
    # the serial reference form is chosen when the batch object is created (measurement build, TOLFG_X0_SERIAL: tol_amd/csrc/knobs.h)
    os.environ["TOLFG_X0_SERIAL"] = "1"
    bs = tol_amd.Batch(mission, air, ts=ts, dtype=dtype)
    os.environ.pop("TOLFG_X0_SERIAL", None)
    bs.set_trajectories(BN.make_trajectories(tol_amd, B, 0, mission, len(air)))
    for obj in (bt, bs):
        for _ in range(3):
            obj.x0_device(dX)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            obj.x0_device(dX)
        e1.record()
        torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    bs.close()
    nbytes = dX.element_size() * B * bt.n
    print(f"{mission:5s} {dtype} B={B:5d} ts={ts:4d}: node-parallel {out[0]:8.1f} us ({nbytes / out[0] / 1e3:6.0f} GB/s written)   serial {out[1]:8.1f} us   first call, host clock {first:7.0f} us", flush=True)
    bt.close()
