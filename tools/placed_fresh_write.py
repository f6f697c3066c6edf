#!/usr/bin/env python3
"""Does a write that follows tolfg_device_alloc at once survive?  (Round 5: tests/test_placed_alloc.py once read 0.0 from a 1-element
block it had just filled with 1.0; torch's caching allocator never shows it.)  Per block size, `n` times: allocate, fill with 1.0,
synchronise, count the elements that are not 1.0 (on the device), look again 5 ms later, free.  Forms: TOLFG_PLACE_SETTLE = 0 (the
block handed out at once: round 4's allocator) and 1 (the block settled first: the shipped form; problem.cpp settle_block) -- measurement
build, tol_amd/csrc/knobs.h -- torch's caching allocator and raw hipMalloc / hipFree for comparison.  (Round 5 also tried waiting for the
chunks' fences through a dma-buf poll: no effect, profiles/r05_fresh_vmm_blocks.md.)  Prints how often writes were lost,
how many elements, which values, where in the block, and what an allocation costs."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tol_amd  # noqa: E402


class _Owned:
    def __init__(self, ptr, count):
        self.ptr = ptr
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def __del__(self):
        tol_amd.capi._hip_runtime.hipFree(C.c_void_p(self.ptr))


def raw_hipmalloc(count):
    p = C.c_void_p()
    hip = tol_amd.capi._hip_runtime
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    assert hip.hipMalloc(C.byref(p), 8 * count) == 0
    return torch.as_tensor(_Owned(p.value, count), device="cuda")


def never_free():
    """`placed_fresh_write.py never-free`: round 4's allocator (TOLFG_PLACE_SETTLE=0), blocks allocated and filled one after the other with
    none freed in between -- does the zeroing follow frees, or every allocation?  (profiles/r05/never_free.txt: it follows frees.)"""
    M = tol_amd.measure_lib()
    os.environ["TOLFG_PLACE_SETTLE"] = "0"
    tol_amd.Batch("S10", ["tempest"], ts=4, library=M).close()
    torch.zeros(1, device="cuda")
    for count in (1 << 20, 32 << 20):
        keep, lost_now, lost_later = [], 0, 0
        for i in range(60 if count == 1 << 20 else 20):
            t = tol_amd.device_alloc((count,), "f64", library=M)
            t.fill_(1.0)
            torch.cuda.synchronize()
            lost_now += int((t != 1.0).sum()) > 0
            keep.append(t)
        time.sleep(0.05)
        for t in keep:
            lost_later += int((t != 1.0).sum()) > 0
        print(f"never freeing, {count} doubles: {lost_now} of {len(keep)} blocks had lost writes right after the fill, {lost_later} when looked at "
              f"50 ms after the last allocation", flush=True)
        del keep, t
        torch.cuda.synchronize()
        time.sleep(0.5)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "never-free":
        return never_free()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    forms = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "torch", "hipMalloc"]
    M = tol_amd.measure_lib()
    torch.zeros(1, device="cuda")
    for count in (3000, 1 << 20, 32 << 20):
        for form in forms:
            if form not in ("torch", "hipMalloc"):
                os.environ["TOLFG_PLACE_SETTLE"] = form
                tol_amd.Batch("S10", ["tempest"], ts=4, library=M).close()      # creating an object makes the measurement build read its variables again
            bad_trials, bad_elems, later_bad, values, where, alloc_s = 0, 0, 0, set(), [], 0.0
            trials = n if count < (32 << 20) else max(n // 4, 20)
            for i in range(trials):
                t0 = time.perf_counter()
                if form == "torch":
                    t = torch.empty(count, dtype=torch.float64, device="cuda")
                elif form == "hipMalloc":          # the runtime's own allocator, no caching in between: hipMalloc ... hipFree every trial
                    t = raw_hipmalloc(count)
                else:
                    t = tol_amd.device_alloc((count,), "f64", library=M)
                alloc_s += time.perf_counter() - t0
                t.fill_(1.0)
                torch.cuda.synchronize()
                wrong = (t != 1.0)
                k = int(wrong.sum())
                if k:
                    bad_trials += 1
                    bad_elems += k
                    idx = wrong.nonzero().flatten()
                    values.update(float(v) for v in t[idx[:4]].tolist())
                    where.append((int(idx[0]), int(idx[-1]), k))
                time.sleep(0.005)
                later_bad += int((t != 1.0).sum()) != k
                del t
            print(f"{count:9d} doubles, settle {form:5s}: {bad_trials:3d} of {trials} allocations lost writes ({bad_elems} elements, values {sorted(values)[:3]}), "
                  f"count changed 5 ms later in {later_bad}; {1e3 * alloc_s / trials:7.2f} ms per allocation; first/last/count: {where[:3]}", flush=True)


if __name__ == "__main__":
    main()
