#!/usr/bin/env python3
"""Do fewer concurrent store fronts rescue a buffer of the slow class?  Eight G buffers side by side, the fastest and the slowest
by the bare store loop of the default plan; on both, the bare store loop and the evaluation with the CU capped at 1 ... 12
resident tile waves (TOLFG_WAVES_PER_CU, read at batch creation)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
trajs = BN.make_trajectories(tol_amd, B, 0, "mixed", 5)


def batch(cap=None):
    if cap is None:
        os.environ.pop("TOLFG_WAVES_PER_CU", None)
    else:
        os.environ["TOLFG_WAVES_PER_CU"] = str(cap)
    bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
    bt.set_trajectories(trajs)
    return bt


base = batch()
dXs, dF, dG0 = BN.make_inputs(base, torch, B, 0, 4)
BN.settle(lambda i: base.eval(dXs[i % 4], dF, dG0, B=B), torch.cuda.synchronize, 5)
bufs = [dG0] + [base.alloc_outputs(B, tries=1) for _ in range(8)]
cls = []
for G in bufs:
    _, st = BN.store_shape_rate(base, torch, dXs, dF, G, B, ts, 104, reps=20)
    cls.append(st)
print("store loop of the default plan on the nine buffers:", [round(c, 1) for c in cls], flush=True)
fast, slow = bufs[cls.index(min(cls))], bufs[cls.index(max(cls))]


def ev_time(bt, G, reps=30):
    for i in range(5):
        bt.eval(dXs[i % 4], dF, G, B=B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        bt.eval(dXs[i % 4], dF, G, B=B)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


print(f"{'cap':>4s} | fast buffer: store loop, evaluation | slow buffer: store loop, evaluation   (us)")
for cap in (1, 2, 3, 4, 5, 6, 8, 10, 12):
    bt = batch(cap)
    row = []
    for G in (fast, slow):
        _, st = BN.store_shape_rate(bt, torch, dXs, dF, G, B, ts, 104, reps=20)
        row.append((st, ev_time(bt, G)))
    print(f"{cap:4d} | {row[0][0]:7.1f} {row[0][1]:7.1f} | {row[1][0]:7.1f} {row[1][1]:7.1f}", flush=True)
    bt.close()
