#!/bin/bash
# what attaching start/stop events to every dispatch costs: bench.py reports the uninstrumented timed region and the instrumented pass
O=gpurun_out/r02ai; mkdir -p $O
for a in "--batch 4096" "--batch 4096" "--batch 1024 --steps 500" "--batch 128 --steps 2000" "--batch 4096 --ts 2000 --aircraft skywalker --batch 400" "--batch 8192 --mission mixed" "--batch 4096 --dtype f32"; do
  timeout -k 10 200 python bench.py --no-configs --no-cpu-baseline $a > $O/b.json 2>/dev/null
  echo "bench.py $a"; python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("  uninstrumented: %.1f us/step wall, %.1f us/launch between two stream events = %.3f of peak | instrumented pass: %.1f us/step wall, dispatch events avg %.1f us (min %.1f) = %.3f of peak"
      % (1e3*d["ms_per_step"], 1e3*r["kernel_ms"], r["frac"], 1e3*r["instrumented_ms_per_step"], 1e3*r["instrumented_kernel_ms"], 1e3*r["instrumented_kernel_min_ms"],
         r["algorithmic_bytes_per_launch"]/(r["instrumented_kernel_ms"]*1e-3)/1e9/r["peak"]))
PY
done 2>&1 | tee $O/event_cost.txt
