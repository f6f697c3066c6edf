#!/bin/bash
# do the XCDs finish their eighths at systematically different times?  (stamped build: timelines only)
O=gpurun_out/r02ad; mkdir -p $O
{
for rep in 1 2 3; do
  for a in "4096 200 20 0 8 1 1 1" "4096 200 20 0 8 0 1 1" "8192 200 10 0 8 1 1 1" "400 2000 20 0 8 1 1 1"; do
    echo "### fgprobe $a   (B N reps variant cap xcd fused nt)"
    timeout -k 10 120 tools/bin/fgprobe $a | grep "per XCC\|us/launch\|resident tile"
  done
done
} > $O/fgprobe.txt 2>&1
echo "exit $?"; cat $O/fgprobe.txt
