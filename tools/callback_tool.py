#!/usr/bin/env python3
"""The SNOPT callback (host x -> host F, G through DEFINEGusrfg_) under the measurement build's switches.

  callback_tool.py rate [reps]   us per call and node-evals/s for a few sizes (bench.callback_mode)
  callback_tool.py ab            x read in place vs always staged (TOLFG_CALLBACK_COPY_X=1), ts = 200 / 2000 / 500
  callback_tool.py trace         where a call spends its time (TOLFG_TRACE=1: one line per call on stderr), zero-copy vs staged

Loads tol_amd/lib/libtolfg_measure.so (tol_amd/csrc/knobs.h) unless TOLFG_LIBRARY says otherwise: the shipped library ignores
every TOLFG_* variable but three."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TOLFG_LIBRARY", os.path.join(ROOT, "tol_amd", "lib", "libtolfg_measure.so"))
mode = sys.argv[1] if len(sys.argv) > 1 else "rate"
if mode == "trace":
    os.environ["TOLFG_TRACE"] = "1"
import bench      # noqa: E402
import tol_amd    # noqa: E402

if mode == "rate":
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        for (m, a, ts, c) in (("S10", "tempest", 100, 500), ("S10", "tempest", 200, 500), ("S10", "tempest", 500, 300),
                              ("S10", "skywalker", 2000, 200), ("G7", "tempest", 100, 500)):
            r = bench.callback_mode(tol_amd, m, a, ts, c)
            print(m, a, ts, "%.1f us/call native (%.1f through ctypes)  %.3g node-evals/s" % (r["us_per_call"], r["us_per_call_via_python_ctypes"], r["node_evals_per_s"]))
elif mode == "ab":
    for copy_x in ("", "1"):
        os.environ.pop("TOLFG_CALLBACK_COPY_X", None)
        if copy_x:
            os.environ["TOLFG_CALLBACK_COPY_X"] = copy_x
        for ts, af in ((200, "tempest"), (2000, "skywalker"), (500, "tempest")):
            p = tol_amd.Problem("S10", af, ts=ts)
            x = p.x0() * 1.001
            us, F, G = p.time_callback(x, 1000, warm=100)
            print("x %s, ts=%d: %.2f us per call" % ("staged" if copy_x else "in place", ts, us))
            p.close()
elif mode == "trace":
    for ts, ac in ((200, "tempest"), (2000, "skywalker")):
        for staging in ("0", "1"):
            os.environ["TOLFG_CALLBACK_STAGING"] = staging
            p = tol_amd.Problem("S10", ac, ts=ts)
            x = p.x0()
            sys.stderr.write(f"--- ts={ts} staging={staging}\n")
            for _ in range(6):
                p.define_fg(x)
            p.close()
else:
    sys.exit(__doc__)
