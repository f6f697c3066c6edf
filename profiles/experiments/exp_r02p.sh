#!/bin/bash
# round-2 experiment P: where a launch's time goes at B = 128 ... 4096 (stamped build: shares and timeline, not benchmark numbers)
mkdir -p gpurun_out/r02p
O=gpurun_out/r02p
{
for cfg in "4096 200 20 0 8 1 1 1" "2048 200 20 0 8 1 1 1" "1024 200 20 0 0 1 1 0" "512 200 20 0 0 1 1 0" "256 200 20 0 0 1 1 0" "128 200 20 0 0 1 1 0" "1 200 20 0 0 1 1 0" "400 2000 20 0 8 1 1 1"; do
echo "### fgprobe $cfg   (B N reps variant cap xcd fused nt)"
timeout -k 5 60 tools/bin/fgprobe $cfg
echo
done
} > $O/fgprobe.txt 2>&1
cat $O/fgprobe.txt
