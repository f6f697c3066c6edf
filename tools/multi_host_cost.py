#!/usr/bin/env python3
"""What a step of the native multi-GPU path costs the HOST when there are P parts to issue for (one device here, P one-rank communicators
of the real library: TOLFG_MULTI_SOLO_COMMS, measurement build): small shards, the slot wait as a stream marker so that the issuing
threads never block -- `issue_us_per_step` is then the host's own time per step: one group bracket around P collective calls from the
caller's thread, against one call per device thread, against no collective (objectives stored to host).  usage: multi_host_cost.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TOLFG_LIBRARY", os.path.join(ROOT, "tol_amd", "lib", "libtolfg_measure.so"))
os.environ["TOLFG_MULTI_SOLO_COMMS"] = "1"
os.environ["TOLFG_MULTI_SLOT_WAIT"] = "stream"
import tol_amd    # noqa: E402
import bench      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for parts in (1, 2, 4, 8):
    for per in (16, 128):
        total = per * parts
        for issue, gather in (("grouped", "rccl"), ("threads", "rccl"), ("grouped", "host"), ("threads", "host")):
            m = tol_amd.Multi("S10", ("tempest",), ts=200, devices=[0] * parts)
            m.set_issue(issue)
            m.set_gather(gather)
            m.set_placement(1)
            m.set_trajectories(bench.make_trajectories(tol_amd, total, 0, "S10", 1))
            m.x0()
            m.sync()
            m.time_steps(20, warm=5)
            t = m.time_steps(steps, warm=20)
            plain = m.time_steps(steps, warm=20, gather=False)
            print(f"{parts} parts x {per:3d} trajectories, issue {issue:7s} gather {gather:4s}: host issue {t['issue_us_per_step']:6.1f} us per step "
                  f"(launches alone {plain['issue_us_per_step']:5.1f}); whole step {t['wall_us_per_step']:6.1f} us (launches alone {plain['wall_us_per_step']:6.1f})", flush=True)
            m.close()
