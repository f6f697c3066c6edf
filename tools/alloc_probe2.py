#!/usr/bin/env python3
"""Follow-up of alloc_probe.py: is the slow class of allocations slow for every way of dealing the tiles to the XCDs?
Six G buffers (torch allocator, kept alive); per launch-plan variant (environment overrides read at batch creation)
the evaluation and the bare store loop on each of them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
trajs = BN.make_trajectories(tol_amd, B, 0, "mixed", 5)


def make(env):
    for k in ("TOLFG_XCD_SHIFT", "TOLFG_XCD", "TOLFG_WAVES_PER_CU", "TOLFG_NT_STORES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
    bt.set_trajectories(trajs)
    return bt


def ev_time(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


base = make({})
dXs, dF, dG0 = BN.make_inputs(base, torch, B, 0, 4)
BN.settle(lambda i: base.eval(dXs[i % 4], dF, dG0, B=B), torch.cuda.synchronize, 5)
Gs = [dG0] + [torch.empty_like(dG0) for _ in range(5)]
variants = [("default (eighths)", {})] + [(f"XCD_SHIFT={s}", {"TOLFG_XCD_SHIFT": str(s)}) for s in (2, 4, 6, 8, 10, 11)] + \
           [("in order (XCD=0)", {"TOLFG_XCD": "0"}), ("cap 4 waves/CU", {"TOLFG_WAVES_PER_CU": "4"}), ("cap 2 waves/CU", {"TOLFG_WAVES_PER_CU": "2"}),
            ("cap 12 waves/CU", {"TOLFG_WAVES_PER_CU": "12"}), ("plain stores", {"TOLFG_NT_STORES": "0"}), ("default again", {})]
for name, env in variants:
    bt = make(env)
    bt.eval(dXs[0], dF, Gs[0], B=B)
    torch.cuda.synchronize()
    row = []
    for G in Gs:
        ev = ev_time(lambda i: bt.eval(dXs[i % 4], dF, G, B=B))
        _, st = BN.store_shape_rate(bt, torch, dXs, dF, G, B, ts, 104, reps=30)
        row.append(f"{ev:6.1f}/{st:6.1f}")
    print(f"{name:20s} eval/store-loop us per buffer: " + "  ".join(row), flush=True)
    bt.close()
