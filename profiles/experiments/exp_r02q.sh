#!/bin/bash
# round-2 experiment Q: occupancy over time inside a launch (stamped build)
mkdir -p gpurun_out/r02q
O=gpurun_out/r02q
{
for cfg in "4096 200 20 0 8 1 1 1" "4096 200 20 0 8 0 1 1" "4096 200 20 0 8 1 0 1" "4096 200 20 0 0 1 1 1" "4096 200 20 0 12 1 1 1" "1024 200 20 0 0 1 1 0"; do
echo "### fgprobe $cfg   (B N reps variant cap xcd fused nt)"
timeout -k 5 60 tools/bin/fgprobe $cfg | grep -v "cycles  "
echo
done
} > $O/fgprobe.txt 2>&1
cat $O/fgprobe.txt
