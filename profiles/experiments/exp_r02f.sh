#!/bin/bash
# round-2 experiment F: GPU suite after the mixed-mission / callback changes, callback rate, mixed + fp32-G7 timings
mkdir -p gpurun_out/r02f
O=gpurun_out/r02f
timeout -k 10 400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest_gpu.log | cut -c1-300
echo "== callback (default)"; timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -5
echo "== callback TOLFG_NO_FLAG"; TOLFG_NO_FLAG=1 timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -5
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  4096,200,64,7,1,1 4096,200,64,7,1,2 8192,200,64,7,1,2 \
  4096,200,64,8,1,1,1 4096,200,64,12,1,1,1 8192,200,64,8,1,2,1 8192,200,64,12,1,2,1 8192,200,64,8,1,0,1 \
  4096,200,64,7,1 4096,200,64,8,1 4096,200,64,7,1 4096,200,64,8,1 4096,200,64,6,1 4096,200,64,9,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
