#!/usr/bin/env python3
"""Callback-mode rate (host x -> host F, G through DEFINEGusrfg_) for a few sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, tol_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for _ in range(reps):
  for (m, a, ts, c) in (("S10", "tempest", 100, 500), ("S10", "tempest", 200, 500), ("S10", "tempest", 500, 300),
                      ("S10", "skywalker", 2000, 200), ("G7", "tempest", 100, 500)):
        r = bench.callback_mode(tol_amd, m, a, ts, c)
        print(m, a, ts, "%.1f us/call native (%.1f through ctypes)  %.3g node-evals/s" % (r["us_per_call"], r["us_per_call_via_python_ctypes"], r["node_evals_per_s"]))
