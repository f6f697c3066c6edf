#!/usr/bin/env python3
"""Fourth allocation probe: at what granularity is a buffer fast or slow?  Several G buffers of the mixed 8192 launch (torch's
allocator and the library's 2 MiB-chunk form); on each the bare store loop of the whole launch and of row windows of
4096 / 2048 / 1024 / 512 rows (non-temporal stores and the cap of 8 forced, as in the whole launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TOLFG_NT_STORES"] = "1"
os.environ["TOLFG_WAVES_PER_CU"] = "8"
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
trajs = BN.make_trajectories(tol_amd, B, 0, "mixed", 5)
full = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
full.set_trajectories(trajs)
subs = {}
for w in (4096, 2048, 1024, 512):
    b = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
    b.set_trajectories(trajs[:w])
    subs[w] = b
dXs, dF, dG0 = BN.make_inputs(full, torch, B, 0, 1)
bufs = [("placed (best of 4)", dG0), ("torch #1", torch.empty_like(dG0)), ("torch #2", torch.empty_like(dG0)),
        ("2 MiB chunks #1", tol_amd.device_alloc(tuple(dG0.shape))), ("2 MiB chunks #2", tol_amd.device_alloc(tuple(dG0.shape))),
        ("torch #3", torch.empty_like(dG0))]
BN.settle(lambda i: full.eval(dXs[0], dF, dG0, B=B), torch.cuda.synchronize, 5)
for name, G in bufs:
    _, whole = BN.store_shape_rate(full, torch, dXs, dF, G, B, ts, 104, reps=20)
    line = f"{name:20s} whole {whole:6.1f} us |"
    for w in (4096, 2048, 1024, 512):
        ts_ = []
        for r0 in range(0, B, w):
            _, us = BN.store_shape_rate(subs[w], torch, [dXs[0][r0:r0 + w]], dF[r0:r0 + w], G[r0:r0 + w], w, ts, 104, reps=20)
            ts_.append(us)
        line += f" {w}: " + " ".join(f"{t:.0f}" for t in ts_) + " |"
    print(line, flush=True)
