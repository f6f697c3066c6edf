#!/bin/bash
# round-2 experiment T: final build -- GPU suite, callback timings incl. F-only calls
mkdir -p gpurun_out/r02t
O=gpurun_out/r02t
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python tools/show_bench.py $O/bench.json
