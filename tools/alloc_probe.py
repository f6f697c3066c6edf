#!/usr/bin/env python3
"""Does the speed of the headline launch depend on WHERE its output buffers were allocated?  (r04: evaluations and the bare
store loop of the same launch fall into two classes, 15 % apart, between allocations of the same size in one process, while
the vendor fill runs at the same speed on all of them.)
One process, the mixed 8192 / ts=200 / fp64 batch.  G buffers from three sources, all kept alive: torch's allocator,
hipMalloc, and hipExtMallocWithFlags(hipDeviceMallocContiguous) = physically contiguous VRAM; on each the vendor fill, the
evaluation, the bare store loop, and the store loop on the first and second half of the rows."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
from tol_amd import capi
import bench as BN

B, ts = 8192, 200
bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, "mixed", 5))
half = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype="f64")
half.set_trajectories(BN.make_trajectories(tol_amd, B // 2, 0, "mixed", 5))
hip = capi._hip_runtime
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]


class Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr, False), "version": 2}


def raw_tensor(shape, contiguous):
    p = C.c_void_p()
    n = 8 * shape[0] * shape[1]
    rc = hip.hipExtMallocWithFlags(C.byref(p), n, 0x4) if contiguous else hip.hipMalloc(C.byref(p), n)
    if rc != 0:
        raise RuntimeError(f"allocation failed with {rc}")
    return torch.as_tensor(Raw(p.value, shape), device="cuda")


def ev_time(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


dXs, dF, dG0 = BN.make_inputs(bt, torch, B, 0, 4)
BN.settle(lambda i: bt.eval(dXs[i % 4], dF, dG0, B=B), torch.cuda.synchronize, 5)
shape = tuple(dG0.shape)
bufs = [("torch", dG0)]
for k in range(3):
    bufs.append(("contiguous", raw_tensor(shape, True)))
    bufs.append(("hipMalloc", raw_tensor(shape, False)))
    bufs.append(("torch", torch.empty_like(dG0)))
for rnd in range(2):
    for kind, G in bufs:
        fill = ev_time(lambda i: G.fill_(1.0 + i), 10)
        ev = ev_time(lambda i: bt.eval(dXs[i % 4], dF, G, B=B))
        _, st = BN.store_shape_rate(bt, torch, dXs, dF, G, B, ts, 104, reps=30)
        _, h0 = BN.store_shape_rate(half, torch, dXs, dF, G, B // 2, ts, 104, reps=30)
        _, h1 = BN.store_shape_rate(half, torch, [dXs[0][B // 2:]], dF[B // 2:], G[B // 2:], B // 2, ts, 104, reps=30)
        print(f"round {rnd} {kind:10s} G @ {G.data_ptr():#x}: fill {fill:6.1f} us  eval {ev:6.1f} us = {bt.algorithmic_bytes(B) / ev / 8e6:5.3f} of peak  "
              f"store loop {st:6.1f} us  rows 0-4095 {h0:6.1f} us  rows 4096-8191 {h1:6.1f} us", flush=True)
