#!/bin/bash
# one gpurun call: parity tests, bench, rocprof profile
TAG=${1:-r01}
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
# the same suite with the single-launch callback kernel disabled: small problems through fg_kernel + finalize_kernel too
TOLFG_NO_SINGLE_LAUNCH=1 timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_twokernel.log 2>&1; echo "pytest (two-kernel path) exit $?"; tail -2 gpurun_out/pytest_gpu_twokernel.log
timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; tail -c 2500 gpurun_out/bench_$TAG.json
timeout -k 10 600 bash tools/profile_gpu.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; echo "profile exit $?"; tail -3 gpurun_out/profile_$TAG.log
