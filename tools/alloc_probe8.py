#!/usr/bin/env python3
"""What distinguishes a slow output buffer: address translation or the DRAM mapping?  Eight G buffers (2 MiB chunks, kept side
by side: on several boxes the first few of a process are fast and the later ones slow); on each the bare store loop of the
headline launch (the class), the vendor fill (a linear sweep), and scattered writes of whole rows in a random order -- rows of
4 KiB, 1 KiB and 128 B through torch's index_copy_.  A buffer whose pages are small or scattered costs a scattered writer
translation misses that a linear sweep never sees; a DRAM-mapping effect would not show in a random order at all."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, "mixed", 5))
dXs, dF, dG0 = BN.make_inputs(bt, torch, B, 0, 4)
BN.settle(lambda i: bt.eval(dXs[i % 4], dF, dG0, B=B), torch.cuda.synchronize, 5)


def ev_time(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


nel = dG0.numel()
scat = {}
for row in (512, 128, 16):          # elements per row: 4 KiB, 1 KiB, 128 B
    n = min(nel // row, (256 << 20) // (8 * row))          # 256 MB of rows per scatter, spread over the whole buffer
    stride = (nel // row) // n
    idx = (torch.randperm(n, device="cuda") * stride).contiguous()
    src = torch.ones((n, row), dtype=torch.float64, device="cuda")
    scat[row] = (idx, src)

bufs = [("torch", dG0)]
for k in range(8):
    bufs.append((f"2 MiB chunks #{k + 1}", bt.alloc_outputs(B, tries=1)))
for tag, G in bufs:
    _, st = BN.store_shape_rate(bt, torch, dXs, dF, G, B, ts, 104, reps=20)
    ev = ev_time(lambda: bt.eval(dXs[0], dF, G, B=B))
    fill = ev_time(lambda: G.fill_(1.0))
    flat = G.view(-1)
    out = []
    for row, (idx, src) in scat.items():
        rows = flat[: (nel // row) * row].view(nel // row, row)
        us = ev_time(lambda: rows.index_copy_(0, idx, src), reps=5)
        out.append(f"{8 * row:5d}-B rows {us:7.1f} us ({src.numel() * 8 / us / 1e3:5.0f} GB/s)")
    print(f"{tag:18s} store loop {st:6.1f} us  eval {ev:6.1f}  fill {fill:6.1f}   scattered: " + "; ".join(out), flush=True)
