#!/bin/bash
# round-2 experiment E: SNOPT-callback latency -- completion word, registered caller arrays, zero-copy limit
mkdir -p gpurun_out/r02e
O=gpurun_out/r02e
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
echo "== default (flag + registered arrays)"; timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -6
echo "== TOLFG_NO_FLAG"; TOLFG_NO_FLAG=1 timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -6
echo "== TOLFG_NO_REGISTER"; TOLFG_NO_REGISTER=1 timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -6
echo "== TOLFG_NO_FLAG TOLFG_NO_REGISTER (round-1 behaviour)"; TOLFG_NO_FLAG=1 TOLFG_NO_REGISTER=1 timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -6
echo "== zero-copy limit 4 MB (ts=2000 direct)"; TOLFG_ZERO_COPY_LIMIT=4000000 timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -6
echo "== zero-copy limit 4 MB, nt stores"; TOLFG_NT_STORES=1 TOLFG_ZERO_COPY_LIMIT=4000000 timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -6
echo "== trace"; timeout -k 10 120 python tools/trace_callback.py > $O/trace.out 2> $O/trace.err; grep -A4 -- "---" $O/trace.err | grep -v amdgpu.ids | head -40
