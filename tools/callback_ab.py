import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tol_amd, numpy as np, sys
for ts, af in ((200, "tempest"), (2000, "skywalker"), (500, "tempest")):
    p = tol_amd.Problem("S10", af, ts=ts)
    x = p.x0() * 1.001
    us, F, G = p.time_callback(x, 1000, warm=100)
    print("ts=%d: %.2f us per call (x at %d mod 16)" % (ts, us, x.ctypes.data % 16))
    p.close()
