#!/bin/bash
# round-3 full pass: GPU suite in its three forms, smoke, bench.py, shape sweep
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
TOLFG_NO_SINGLE_LAUNCH=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_tiled_callback.log 2>&1; echo "pytest (callback through the tile-per-workgroup path) exit $?"; tail -2 $O/pytest_gpu_tiled_callback.log
TOLFG_FUSED=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_two_launch.log 2>&1; echo "pytest (two-launch form) exit $?"; tail -2 $O/pytest_gpu_two_launch.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python tools/show_bench.py $O/bench.json
timeout -k 10 600 bash tools/shape_sweep.sh > $O/shape_sweep.md 2>&1; echo "shape sweep exit $?"; cat $O/shape_sweep.md
