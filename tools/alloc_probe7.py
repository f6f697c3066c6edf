#!/usr/bin/env python3
"""How many placement candidates are worth holding?  The headline batch's G buffer through tolfg_batch_alloc_outputs with the cap
raised to 16 and the early accept off (TOLFG_PLACE_CAP, TOLFG_PLACE_EARLY): every candidate's bare-store-loop time, the wall time of
the search, and the evaluation on the buffer kept."""
import os
import sys
import time

os.environ["TOLFG_PLACE_CAP"] = "16"
os.environ["TOLFG_PLACE_EARLY"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
for rep in range(3):
    bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
    bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, "mixed", 5))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    G = bt.alloc_outputs(B, tries=16)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    pr = bt.placement["probe_us"]
    best_of = [min(pr[:k]) for k in (1, 2, 4, 6, 8, 12, 16) if k <= len(pr)]
    print(f"rep {rep}: {len(pr)} candidates in {wall:.2f} s: {pr}", flush=True)
    print(f"        best of the first 1/2/4/6/8/12/16: {best_of}", flush=True)
    del G
    bt.close()
