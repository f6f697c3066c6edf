#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters...>" ["bench args"]   (run on the GPU box, repo root)
# One rocprofv3 counter pass (kernel-trace only) over bench.py; prints the fg_kernel averages.
# The profiler is bounded: a counter name gfx950 does not expose makes rocprofv3 abort the profiled
# program at its first dispatch (rocprofiler_create_counter_config error 38, signal 6) and then sit in
# its finalizer -- that is what happened to round 1's TA pass (TA_BUSY_sum does not exist on gfx950).
set -e
TAG=$1; CTRS=$2
ARGS=${3:-"--steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-native-multi"}
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python3 bench.py $ARGS > "$OUT/bench.log" 2>&1 || { echo "rocprofv3 pass $TAG failed or timed out:"; grep -m3 -E "Missing|error|fatal" "$OUT/bench.log"; tail -5 "$OUT/bench.log"; exit 1; }
python3 tools/parse_rocprof.py --avg "$OUT/pmc_counter_collection.csv"
