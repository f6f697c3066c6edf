#!/bin/bash
# The round's full GPU pass, in parts that each fit one gpurun call (<= 20 min):  tools/gpu_run_all.sh <tag> suites | bench | profile
#   suites   GPU suite against the shipped library, then against the measurement build (tol_amd/csrc/knobs.h): default plan, the
#            callback through the one-workgroup kernel, the two-launch form; smoke
#   bench    bench.py as the driver starts it, at its own defaults, as `--gpus 2` plain command (two ranks on this box's one GPU, gloo),
#            and as ONE RCCL rank (--single-rank-collectives): every line carries the native C++ leg (native_multi)
#   profile  rocprofv3 --kernel-trace --stats + PMC passes (tools/profile_gpu.sh; then tools/parse_rocprof.py <tag> 8192 200 f64 mixed), shape sweep
# Raw output under gpurun_out/<tag>/.
TAG=${1:-r05}
PART=${2:-suites}
M=$PWD/tol_amd/lib/libtolfg_measure.so
O=gpurun_out/$TAG
mkdir -p $O
case "$PART" in
suites)
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
TOLFG_LIBRARY=$M timeout -k 10 400 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_spawn.py > $O/pytest_gpu_measure.log 2>&1; echo "pytest (measurement build) exit $?"; tail -2 $O/pytest_gpu_measure.log
TOLFG_LIBRARY=$M TOLFG_FORCE_SINGLE_LAUNCH=1 timeout -k 10 400 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_spawn.py > $O/pytest_gpu_single_callback.log 2>&1; echo "pytest (callback through the one-workgroup kernel) exit $?"; tail -2 $O/pytest_gpu_single_callback.log
TOLFG_LIBRARY=$M TOLFG_FUSED=0 timeout -k 10 400 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_spawn.py > $O/pytest_gpu_two_launch.log 2>&1; echo "pytest (two-launch form) exit $?"; tail -2 $O/pytest_gpu_two_launch.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
;;
bench)
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench (the driver's command) exit $?"; python tools/show_bench.py $O/bench_driver_cmd.json
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python tools/show_bench.py $O/bench.json
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; echo "bench --gpus 2 (plain command, gloo rehearsal) exit $?"; python tools/show_bench.py $O/bench_2ranks_gloo.json | cut -c1-400
timeout -k 10 600 python3 bench.py --gpus 1 --single-rank-collectives --steps 20 --warmup 5 > $O/bench_1rank_rccl.json 2> $O/bench_1rank_rccl.err; echo "bench, one RCCL rank exit $?"; python tools/show_bench.py $O/bench_1rank_rccl.json | cut -c1-400
;;
profile)
timeout -k 10 600 bash tools/profile_gpu.sh $TAG > $O/profile.log 2>&1; echo "profile exit $?"; tail -2 $O/profile.log      # then: python tools/parse_rocprof.py $TAG 8192 200 f64 mixed
timeout -k 10 600 bash tools/shape_sweep.sh > $O/shape_sweep.md 2>&1; echo "shape sweep exit $?"; cat $O/shape_sweep.md
;;
esac
