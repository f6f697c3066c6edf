#!/usr/bin/env python3
"""Where a DEFINEGusrfg_ call spends its time (TOLFG_TRACE=1 prints one line per call on stderr)."""
import os, sys
os.environ["TOLFG_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tol_amd
for ts, ac in ((200, "tempest"), (2000, "skywalker")):
    for mode in ("0", "1"):
        os.environ["TOLFG_CALLBACK_STAGING"] = mode
        p = tol_amd.Problem("S10", ac, ts=ts)
        x = p.x0()
        sys.stderr.write(f"--- ts={ts} staging={mode}\n")
        for _ in range(6):
            p.define_fg(x)
        p.close()
