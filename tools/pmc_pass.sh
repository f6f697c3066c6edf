#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters...>" ["bench args"]   (run on the GPU box, repo root)
# One rocprofv3 counter pass (kernel-trace only) over bench.py; prints the fg_kernel averages.
set -e
TAG=$1; CTRS=$2
ARGS=${3:-"--steps 20 --warmup 3 --no-cpu-baseline --no-callback"}
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python3 bench.py $ARGS > "$OUT/bench.log" 2>&1 || { tail -20 "$OUT/bench.log"; exit 1; }
python3 tools/pmc_avg.py "$OUT/pmc_counter_collection.csv"
