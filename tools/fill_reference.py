#!/usr/bin/env python3
"""Calibration: what the vendor's own fill / copy kernels reach on this GPU for the sizes the fg kernel writes
(torch.Tensor.fill_ / zero_ = hipMemset-class kernels; copy_ = read + write)."""
import time
import torch

def bench(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

for mb in (200, 800, 3200):
    n = mb * 1000 * 1000 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    b = torch.empty(n, dtype=torch.float64, device="cuda")
    t = bench(lambda: a.fill_(1.5))
    print("fill_  %5d MB: %7.1f us  %6.0f GB/s written" % (mb, 1e6 * t, 8 * n / t / 1e9))
    t = bench(lambda: a.zero_())
    print("zero_  %5d MB: %7.1f us  %6.0f GB/s written" % (mb, 1e6 * t, 8 * n / t / 1e9))
    t = bench(lambda: b.copy_(a))
    print("copy_  %5d MB: %7.1f us  %6.0f GB/s written, %6.0f GB/s read + written" % (mb, 1e6 * t, 8 * n / t / 1e9, 16 * n / t / 1e9))
    del a, b
    torch.cuda.empty_cache()
