#!/bin/bash
# round-2 experiment W: bench.py with the two-streams side record
mkdir -p gpurun_out/r02w
timeout -k 10 400 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r02w/bench.json 2> gpurun_out/r02w/bench.err; echo "bench exit $?"; tail -3 gpurun_out/r02w/bench.err; python tools/show_bench.py gpurun_out/r02w/bench.json | tail -4
