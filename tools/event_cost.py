#!/usr/bin/env python3
"""What does the marker between two launches cost that a collective on another stream needs?  The headline batch, launches back to
back on one stream: (a) nothing in between; (b) a non-timing event recorded after every launch; (c) + another stream waiting for
it; (d) + a small kernel on that stream (what the all-gather of the objectives amounts to for the launch stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = (int(sys.argv[1]) if len(sys.argv) > 1 else 8192), 200
bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, "mixed", 5))
dXs, dF, dG = BN.make_inputs(bt, torch, B, 0, 4)
obj = torch.zeros(B, dtype=torch.float64, device="cuda")
out = torch.zeros(B, dtype=torch.float64, device="cuda")
BN.settle(lambda i: bt.eval(dXs[i % 4], dF, dG, obj=obj, B=B), torch.cuda.synchronize, 5)
side = torch.cuda.Stream()
hi = torch.cuda.Stream(priority=-1)


def run(kind, reps=100, s2=side):
    evs = [torch.cuda.Event(enable_timing=False) for _ in range(reps)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        bt.eval(dXs[i % 4], dF, dG, obj=obj, B=B)
        if kind >= 1:
            evs[i].record()
        if kind >= 2:
            s2.wait_event(evs[i])
        if kind >= 3:
            with torch.cuda.stream(s2):
                out.copy_(obj)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


names = ["(a) launches back to back", "(b) + event recorded after every launch", "(c) + another stream waits for it", "(d) + a small kernel on that stream",
         "(d) with a high-priority stream"]
for rnd in range(3):
    row = [run(0), run(1), run(2), run(3), run(3, s2=hi)]
    print(f"B={B} round {rnd}: " + "; ".join(f"{n} {t:.1f} us" for n, t in zip(names, row)), flush=True)
