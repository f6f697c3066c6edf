#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root.  Writes raw rocprofv3 output under
# gpurun_out/prof_<tag>/ ; tools/parse_rocprof.py turns it into the committed profiles/ summaries.
# Counters are collected in their own passes (kernel-trace only, no other trace domains).
set -e
TAG=${1:-r01}
# (--no-native-multi: under the profiler bench.py must not start its native-leg child -- the profiler's preloaded library would ride into it)
ARGS=${2:-"--steps 50 --warmup 5 --no-cpu-baseline --no-configs --no-calibration --no-native-multi"}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "[profile] the same command without the profiler (same box, same call): its own HIP-event figure goes beside the profiler's"
python3 bench.py ${ARGS/--no-calibration/} > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
echo "[profile] kernel trace + stats"
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 bench.py $ARGS > "$OUT/bench_stats.log" 2>&1
echo "[profile] pmc FETCH_SIZE"
timeout -k 5 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 bench.py $ARGS > "$OUT/bench_fetch.log" 2>&1
echo "[profile] pmc WRITE_SIZE"
timeout -k 5 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 bench.py $ARGS > "$OUT/bench_write.log" 2>&1
find "$OUT" -name "*.csv" | head -50
# keep the merged payload small: kernel-trace CSVs of the bench are a few hundred rows
du -sh "$OUT"
