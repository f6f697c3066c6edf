#!/usr/bin/env python3
"""Fifth allocation probe: does sliding the G buffer inside a larger allocation, or padding its row stride, change the class?
One 2.6 GB torch allocation and one of the library's (2 MiB chunks); G as a view at offsets 0 ... 1024 MiB; then row strides
padded by 0 ... 4096 elements at offset 0.  Bare store loop of the mixed 8192 launch and the evaluation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

B, ts = 8192, 200
bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, "mixed", 5))
dXs, dF, dG0 = BN.make_inputs(bt, torch, B, 0, 2)
ld = dG0.shape[1]
BN.settle(lambda i: bt.eval(dXs[i % 2], dF, dG0, B=B), torch.cuda.synchronize, 5)
print("placement of the library's own buffer:", bt.placement["probe_us"], flush=True)


def ev_time(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def measure(tag, G):
    _, st = BN.store_shape_rate(bt, torch, dXs, dF, G, B, ts, 104, reps=20)
    ev = ev_time(lambda i: bt.eval(dXs[i % 2], dF, G, B=B))
    print(f"{tag:46s} store loop {st:6.1f} us   eval {ev:6.1f} us", flush=True)


n = B * ld
extra = (1024 << 20) // 8 + B * 4096
for kind in ("torch", "2 MiB chunks"):
    big = torch.empty(n + extra, dtype=torch.float64, device="cuda") if kind == "torch" else tol_amd.device_alloc((n + extra,))
    for mib in (0, 2, 6, 16, 50, 128, 300, 512, 777, 1024):
        off = (mib << 20) // 8
        measure(f"{kind}: G at +{mib} MiB", big[off:off + n].view(B, ld))
    for pad in (2, 16, 128, 1024, 4096):
        measure(f"{kind}: row stride + {pad} elements", big[:B * (ld + pad)].view(B, ld + pad))
    del big
    torch.cuda.empty_cache()
