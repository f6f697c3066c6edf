#!/usr/bin/env python3
"""ONE experiment on round 5's finding that the placement class of the 1.4 GB output buffer is a state of the device's memory
(profiles/r05_placement_state.md): on a device that offers only slow candidates, does WHERE in the device's memory the candidates are
taken from matter?  Six candidates per search (early accept off: measurement build), the search repeated while dummy blocks of 64 / 128 /
200 GB are held (pushing the candidates elsewhere in the physical memory), after they are freed again, and after a pause.
Kill criterion: no search under any condition finds a candidate >= 15 % faster than the baseline search's best -> no recipe, stop."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TOLFG_LIBRARY", os.path.join(ROOT, "tol_amd", "lib", "libtolfg_measure.so"))
os.environ["TOLFG_PLACE_EARLY"] = "0"
import torch      # noqa: E402
import bench      # noqa: E402
import tol_amd    # noqa: E402

B = 8192
bt = tol_amd.Batch("mixed", bench.AIRCRAFT5, ts=200, dtype="f64")
bt.set_trajectories(bench.make_trajectories(tol_amd, B, 0, "mixed", 5))


def search(tag):
    t0 = time.perf_counter()
    G = bt.alloc_outputs(B, tries=6)
    pr = bt.placement["probe_us"]
    free_b, total_b = torch.cuda.mem_get_info()
    print(f"{tag:46s}: best {min(pr):6.1f}  all {pr}  ({time.perf_counter() - t0:4.1f} s; {free_b / 2**30:5.1f} GiB free of {total_b / 2**30:5.1f})", flush=True)
    del G
    return min(pr)


base = search("baseline")
search("baseline again")
held = []
for gib in (64, 64, 72):
    held.append(torch.empty(gib << 30, dtype=torch.uint8, device="cuda"))
    held[-1][::1 << 20].fill_(1)
    torch.cuda.synchronize()
    search(f"holding {sum(h.numel() for h in held) >> 30} GiB of plain allocations")
del held
torch.cuda.empty_cache()
torch.cuda.synchronize()
search("after freeing them")
time.sleep(20)
search("20 s later")
# the other end: candidates taken while most of the memory is held by 2 MiB-chunk blocks of the library's own allocator
blocks = [tol_amd.device_alloc((8 << 30,), "f32") for _ in range(3)]      # 3 x 32 GiB
search("holding 96 GiB of the library's chunked blocks")
del blocks
search("after freeing those")
print(f"baseline best {base:.1f} us: a recipe would have to show <= {0.85 * base:.1f} us")
