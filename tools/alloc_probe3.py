#!/usr/bin/env python3
"""Third allocation probe: does the SIZE of the physical blocks behind the output buffer decide which class it falls into?
G buffers built with the HIP virtual-memory API -- one reserved address range, backed by separately created physical chunks of
2 MiB ... 512 MiB (hipMemCreate + hipMemMap) -- next to torch's allocator; on each: vendor fill, evaluation, the bare store loop
(whole launch, first / second half of the rows)."""
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


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle", C.c_int), ("loc", Loc), ("win32", C.c_void_p), ("compression", C.c_ubyte), ("rdma", C.c_ubyte),
                ("usage", C.c_ushort)]


class Access(C.Structure):
    _fields_ = [("loc", Loc), ("flags", C.c_int)]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with {rc}")


def vmm_tensor(shape, chunk, order="sequential", seed=0):
    """order: how the physical chunks (created one after the other) are laid into the address range: "sequential", "shuffled"
    (a random permutation) or "strided:K" (chunk j goes to slot j * K mod n, K coprime to n)."""
    nbytes = 8 * shape[0] * shape[1]
    prop = Prop()
    prop.type, prop.loc.type, prop.loc.id = 1, 1, 0          # pinned, device 0
    gran = C.c_size_t()
    ck(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "hipMemGetAllocationGranularity")
    chunk = max(chunk, gran.value) // gran.value * gran.value
    total = (nbytes + chunk - 1) // chunk * chunk
    ptr = C.c_void_p()
    ck(hip.hipMemAddressReserve(C.byref(ptr), C.c_size_t(total), C.c_size_t(0), None, C.c_ulonglong(0)), "hipMemAddressReserve")
    n = total // chunk
    slots = list(range(n))
    if order == "shuffled":
        import random
        random.Random(seed).shuffle(slots)
    elif order.startswith("strided:"):
        import math
        k = int(order.split(":")[1])
        while math.gcd(k, n) != 1:
            k += 1
        slots = [(j * k) % n for j in range(n)]
    for j in range(n):
        h = C.c_void_p()
        ck(hip.hipMemCreate(C.byref(h), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
        ck(hip.hipMemMap(C.c_void_p(ptr.value + slots[j] * chunk), C.c_size_t(chunk), C.c_size_t(0), h, C.c_ulonglong(0)), "hipMemMap")
    acc = Access()
    acc.loc.type, acc.loc.id, acc.flags = 1, 0, 3
    ck(hip.hipMemSetAccess(ptr, C.c_size_t(total), C.byref(acc), C.c_size_t(1)), "hipMemSetAccess")

    class Raw:
        __cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr.value, False), "version": 2}
    return torch.as_tensor(Raw(), device="cuda"), gran.value


def ev_time(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def measure(tag, btx, dXs, F, G, Bx, slab=104):
    G.zero_()
    ev = ev_time(lambda i: btx.eval(dXs[i % len(dXs)], F, G, B=Bx))
    _, st = BN.store_shape_rate(btx, torch, dXs, F, G, Bx, ts, slab, reps=30)
    print(f"{tag:44s} eval {ev:7.1f} us = {btx.algorithmic_bytes(Bx) / ev / 8e6:5.3f} of peak   store loop {st:7.1f} us", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "order":
    # the same 2 MiB chunks laid into the address range in order, shuffled, or strided: several of each, interleaved
    dXs, dF, dG0 = BN.make_inputs(bt, torch, B, 0, 4)
    BN.settle(lambda i: bt.eval(dXs[i % 4], dF, dG0, B=B), torch.cuda.synchronize, 5)
    shape = tuple(dG0.shape)
    measure("torch", bt, dXs, dF, dG0, B)
    for rnd in range(4):
        for order in ("sequential", "shuffled", "strided:7", "strided:97"):
            g, _ = vmm_tensor(shape, 2 << 20, order, seed=rnd)
            measure(f"2 MiB chunks, {order} #{rnd + 1}", bt, dXs, dF, g, B)
    for mib in (4, 8, 32):
        for rnd in range(2):
            g, _ = vmm_tensor(shape, mib << 20, "shuffled", seed=rnd)
            measure(f"{mib} MiB chunks, shuffled #{rnd + 1}", bt, dXs, dF, g, B)
    g, _ = vmm_tensor(shape, 2 << 20, "shuffled", seed=11)
    f, _ = vmm_tensor(tuple(dF.shape), 2 << 20, "shuffled", seed=12)
    measure("G and F in shuffled 2 MiB chunks", bt, dXs, f, g, B)
    xs = []
    for i, x in enumerate(dXs):
        xv, _ = vmm_tensor(tuple(x.shape), 2 << 20, "shuffled", seed=20 + i)
        xv.copy_(x)
        xs.append(xv)
    measure("G, F and the four X in shuffled 2 MiB chunks", bt, xs, f, g, B)
elif len(sys.argv) > 1 and sys.argv[1] == "sizes":
    # chunk size sweep on the headline launch
    dXs, dF, dG0 = BN.make_inputs(bt, torch, B, 0, 4)
    BN.settle(lambda i: bt.eval(dXs[i % 4], dF, dG0, B=B), torch.cuda.synchronize, 5)
    shape = tuple(dG0.shape)
    measure("torch", bt, dXs, dF, dG0, B)
    for kib in (64, 256, 1024, 2048, 4096, 8192):
        for k in range(2):
            g, gran = vmm_tensor(shape, kib << 10)
            measure(f"G in {kib} KiB chunks #{k + 1}", bt, dXs, dF, g, B)
    # F and X in chunks too
    g, _ = vmm_tensor(shape, 2 << 20)
    f, _ = vmm_tensor(tuple(dF.shape), 2 << 20)
    measure("G and F in 2 MiB chunks", bt, dXs, f, g, B)
    xs = []
    for x in dXs:
        xv, _ = vmm_tensor(tuple(x.shape), 2 << 20)
        xv.copy_(x)
        xs.append(xv)
    measure("G, F and the four X in 2 MiB chunks", bt, xs, f, g, B)
else:
    # other launches: torch's allocation against 2 MiB chunks
    for mission, dtype, Bx, pattern in (("mixed", "f32", 8192, "reference"), ("S10", "f64", 4096, "reference"), ("S10", "f64", 4096, "compact"),
                                        ("S10", "f64", 2048, "reference"), ("S10", "f64", 1024, "reference"), ("mixed", "f32", 1024, "reference"),
                                        ("S10", "f64", 128, "reference")):
        air = BN.AIRCRAFT5 if mission == "mixed" else ("tempest",)
        b2 = tol_amd.Batch(mission, air, ts=ts, dtype=dtype, pattern=pattern)
        b2.set_trajectories(BN.make_trajectories(tol_amd, Bx, 0, mission, len(air)))
        dXs, dF, dG = BN.make_inputs(b2, torch, Bx, 0, 4)
        BN.settle(lambda i: b2.eval(dXs[i % 4], dF, dG, B=Bx), torch.cuda.synchronize, 5)
        tag = f"{mission} {dtype} {Bx} {pattern}"
        slab = 46 if pattern == "compact" else 104
        es = dG.element_size()
        for rnd in range(2):
            measure(tag + ": torch", b2, dXs, dF, dG, Bx, slab)
            if es == 8:
                g, _ = vmm_tensor(tuple(dG.shape), 2 << 20)
                f, _ = vmm_tensor(tuple(dF.shape), 2 << 20)
            else:      # the helper builds float64 views: reinterpret
                g64, _ = vmm_tensor((dG.shape[0], (dG.shape[1] + 1) // 2), 2 << 20)
                f64, _ = vmm_tensor((dF.shape[0], (dF.shape[1] + 1) // 2), 2 << 20)
                g = g64.view(torch.float32)[:, :dG.shape[1]] if dG.shape[1] % 2 == 0 else None
                f = f64.view(torch.float32)[:, :dF.shape[1]] if dF.shape[1] % 2 == 0 else None
            if g is not None and f is not None and g.stride(0) == dG.stride(0) and f.stride(0) == dF.stride(0):
                measure(tag + ": 2 MiB chunks", b2, dXs, f, g, Bx, slab)
        b2.close()
