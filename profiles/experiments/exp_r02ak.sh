#!/bin/bash
# phase shares and timeline of the compact pattern (stamped build: shares only) next to the reference pattern
O=gpurun_out/r02ak; mkdir -p $O
{
for a in "4096 200 20 0 8 1 1 1 0" "4096 200 20 0 8 1 0 1 1" "4096 200 20 0 0 1 0 1 1" "4096 200 20 0 8 1 1 1 1" "4096 200 20 256 8 1 0 1 1" "4096 200 20 2048 8 1 0 1 1"; do
  echo "### fgprobe $a   (B N reps variant cap xcd fused nt pattern)"
  timeout -k 10 120 tools/bin/fgprobe $a
done
} > $O/fgprobe.txt 2>&1
cat $O/fgprobe.txt
