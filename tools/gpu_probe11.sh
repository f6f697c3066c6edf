#!/bin/bash
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import sys, time, os
sys.path.insert(0,'.')
import bench, tol_amd
for mode in ("0","1"):
    os.environ["TOLFG_CALLBACK_STAGING"]=mode
    for (m,a,ts,c) in (("S10","tempest",200,500),("S10","skywalker",2000,200),("G7","tempest",100,500)):
        r=bench.callback_mode(tol_amd,m,a,ts,c)
        print("staging" if mode=="1" else "zero-copy", m,a,ts, "%.1f us/call  %.3g node-evals/s"%(r["us_per_call"], r["node_evals_per_s"]))
PY
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --no-callback 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32', d['value'], d['ms_per_step'], d['roofline'])"
timeout -k 10 600 python tools/fp32_sweep.py > gpurun_out/fp32_sweep.md 2>&1; tail -30 gpurun_out/fp32_sweep.md
