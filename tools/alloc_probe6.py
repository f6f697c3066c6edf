#!/usr/bin/env python3
"""Sixth allocation probe: the library's allocator with chunk sizes other than 2 MiB (TOLFG_PLACED_CHUNK_KIB), several buffers
of each kept side by side: evaluation and bare store loop of the mixed 8192 launch, fp64 and fp32."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
for dtype in ("f64", "f32"):
    bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype=dtype)
    bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, "mixed", 5))
    os.environ.pop("TOLFG_PLACED_CHUNK_KIB", None)
    dXs, dF, dG0 = BN.make_inputs(bt, torch, B, 0, 2)
    BN.settle(lambda i: bt.eval(dXs[i % 2], dF, dG0, B=B), torch.cuda.synchronize, 5)
    print(dtype, "placement of the library's own buffer:", bt.placement["probe_us"], flush=True)
    bufs = [("library (best of 6)", dG0), ("torch", torch.empty_like(dG0))]
    for kib in (2048, 1024, 512, 4096, 2048, 1024, 512):
        os.environ["TOLFG_PLACED_CHUNK_KIB"] = str(kib)
        bufs.append((f"{kib} KiB chunks", tol_amd.device_alloc(tuple(dG0.shape), dtype)))
    os.environ.pop("TOLFG_PLACED_CHUNK_KIB", None)
    for name, G in bufs:
        _, st = BN.store_shape_rate(bt, torch, dXs, dF, G, B, ts, 104, reps=20)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            bt.eval(dXs[i % 2], dF, G, B=B)
        e1.record()
        torch.cuda.synchronize()
        ev = 1e3 * e0.elapsed_time(e1) / 30
        print(f"{dtype} {name:22s} store loop {st:6.1f} us   eval {ev:6.1f} us = {bt.algorithmic_bytes(B) / ev / 8e6:5.3f} of peak", flush=True)
    del bufs, dXs, dF, dG0
    bt.close()
    torch.cuda.empty_cache()
