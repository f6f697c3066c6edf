#!/usr/bin/env python3
"""Experiment: store tokens (FgArgs::store_tokens, TOLFG_STORE_TOKENS=K -- at most K tile waves of a CU stream their slabs at a
time) x resident-wave cap, on the fastest and the slowest of nine G buffers; outputs compared bitwise with the default plan's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
mission, dtype = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("mixed", "f64")
air = BN.AIRCRAFT5 if mission == "mixed" else ("tempest",)
trajs = BN.make_trajectories(tol_amd, B, 0, mission, len(air))


def batch(cap=None, tokens=0):
    for k, v in (("TOLFG_WAVES_PER_CU", cap), ("TOLFG_STORE_TOKENS", tokens or None)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    bt = tol_amd.Batch(mission, air, ts=ts, dtype=dtype)
    bt.set_trajectories(trajs)
    return bt


base = batch()
dXs, dF, dG0 = BN.make_inputs(base, torch, B, 0, 4)
BN.settle(lambda i: base.eval(dXs[i % 4], dF, dG0, B=B), torch.cuda.synchronize, 5)
bufs = [dG0] + [base.alloc_outputs(B, tries=1) for _ in range(8)]
slab = 104
cls = [BN.store_shape_rate(base, torch, dXs, dF, G, B, ts, slab, reps=20)[1] for G in bufs]
print("store loop of the default plan on the nine buffers:", [round(c, 1) for c in cls], flush=True)
fast, slow = bufs[cls.index(min(cls))], bufs[cls.index(max(cls))]
dF.zero_(); fast.zero_()
base.eval(dXs[0], dF, fast, B=B)
torch.cuda.synchronize()
refF, refG = dF.clone(), fast.clone()


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


print(f"{'cap':>4s} {'K':>3s} | evaluation on the fast buffer | on the slow buffer (us)   outputs")
for cap in (None, 12, 16):
    for K in (0, 1, 2, 3, 4, 6):
        bt = batch(cap, K)
        dF.zero_(); fast.zero_()
        bt.eval(dXs[0], dF, fast, B=B)
        torch.cuda.synchronize()
        same = torch.equal(dF, refF) and torch.equal(fast, refG)
        tf = [ev_time(bt, fast) for _ in range(2)]
        tsl = [ev_time(bt, slow) for _ in range(2)]
        print(f"{str(cap):>4s} {K:3d} | {tf[0]:7.1f} {tf[1]:7.1f} | {tsl[0]:7.1f} {tsl[1]:7.1f}   {'bitwise equal' if same else 'DIFFER'}", flush=True)
        bt.close()
