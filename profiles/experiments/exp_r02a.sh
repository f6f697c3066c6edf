#!/bin/bash
# round-2 experiment A (one gpurun call): write-stream shapes, tile size x cap x fused sweep, callback trace
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
echo "== wrbench (mode 4: nt 16-B stores, S KiB sequential per wave)" > $O/wrbench.txt
for lds in 0 23400; do for S in 2 4 8 13 16 26 43; do timeout -k 5 60 tools/bin/wrbench 4 $S $lds >> $O/wrbench.txt 2>&1; done; done
echo "== wrbench mode 6 (read 1 KiB + S KiB nt stores per wave)" >> $O/wrbench.txt
for lds in 0 23400; do for S in 4 7 13; do timeout -k 5 60 tools/bin/wrbench 6 $S $lds 8 1 >> $O/wrbench.txt 2>&1; done; done
cat $O/wrbench.txt
echo "== fgbench"
timeout -k 10 400 tools/bin/fgbench reps=40 \
  4096,200,64,7,0 4096,200,64,7,1 4096,200,64,0,1 4096,200,64,8,1 4096,200,64,6,1 \
  4096,200,48,0,1 4096,200,48,8,1 4096,200,48,10,1 \
  4096,200,32,0,1 4096,200,32,8,1 4096,200,32,10,1 4096,200,32,12,1 4096,200,32,14,1 \
  4096,200,16,0,1 4096,200,16,8,1 4096,200,16,12,1 4096,200,16,16,1 4096,200,16,20,1 \
  4096,200,8,0,1 4096,200,8,16,1 \
  1024,200,64,0,0 1024,200,64,0,1 1024,200,64,7,1 1024,200,32,0,1 1024,200,32,12,1 1024,200,16,0,1 1024,200,16,16,1 1024,200,8,0,1 \
  512,200,64,0,0 512,200,64,0,1 512,200,32,0,1 512,200,16,0,1 512,200,8,0,1 \
  128,200,64,0,0 128,200,64,0,1 128,200,32,0,1 128,200,16,0,1 128,200,8,0,1 \
  400,2000,64,7,0 400,2000,64,7,1 400,2000,32,10,1 400,2000,16,16,1 \
  4096,200,64,7,0,1 4096,200,64,7,1,1 4096,200,16,16,1,1 \
  4096,200,64,8,0,0,1 4096,200,64,8,1,0,1 4096,200,32,12,1,0,1 4096,200,16,16,1,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
echo "== callback trace"
timeout -k 10 120 python tools/trace_callback.py > $O/trace.out 2> $O/trace.err; echo "trace exit $?"; grep -A6 -- "---" $O/trace.err | head -40
echo "== gpu tests"
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
