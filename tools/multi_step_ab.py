#!/usr/bin/env python3
"""What the asynchronous objective gather costs a step of the native multi-GPU host path (tolfg_multi, one device here): the same object,
the same buffers, the native step loop with and without the gather (bench.native_workload), under the measurement build's switch for
the gather stream's priority and both ways of issuing the collective.  usage: multi_step_ab.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TOLFG_LIBRARY", os.path.join(ROOT, "tol_amd", "lib", "libtolfg_measure.so"))
import torch      # noqa: E402
import bench      # noqa: E402
import tol_amd    # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for what, mission, air, total, dtype in (("mixed 8192 f64", "mixed", bench.AIRCRAFT5, 8192, "f64"), ("S10 1024 f64", "S10", ("tempest",), 1024, "f64"),
                                         ("S10 128 f64", "S10", ("tempest",), 128, "f64")):
    for prio, issue, gather, slot_wait in (("1", "grouped", "rccl", "host"), ("1", "grouped", "rccl", "stream"), ("1", "threads", "rccl", "host"), ("1", "threads", "rccl", "stream"),
                                           ("0", "grouped", "rccl", "host"), ("1", "grouped", "host", "host"), ("1", "threads", "host", "host")):
        if True:
            os.environ["TOLFG_MULTI_GATHER_PRIORITY"] = prio
            os.environ["TOLFG_MULTI_SLOT_WAIT"] = slot_wait
            r = bench.native_workload(tol_amd, torch, [0], mission, air, 200, dtype, total, steps, 10, 4, issue, "weak", what, gather=gather)
            print(f"{what:15s} gather {gather:4s} (stream priority {'high' if prio == '1' else 'low '}) issue {issue:7s} slot wait on {slot_wait:6s}: step {1e3 * r['ms_per_step']:7.1f} us with the gather, "
                  f"{1e3 * r['ms_per_step_without_gather']:7.1f} without (launch stream: {r['eval_us']:7.1f} / {r['eval_us_without_gather']:7.1f} us per launch); "
                  f"host issue {r['issue_us_per_step']:5.1f} us per step; one synchronous gather {r['gather_us']:5.1f} us", flush=True)
