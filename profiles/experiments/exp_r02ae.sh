#!/bin/bash
# is the odd/even XCD asymmetry a property of the XCD or of the eighth of the output it walks?  (stamped build: timelines only)
# variant 65536: XCD x walks eighth x^1;  131072: (x+4)%8;  262144: (x+2)%8
O=gpurun_out/r02ae; mkdir -p $O
{
for shape in "4096 200 20" "400 2000 20"; do
  for v in 0 65536 131072 262144 0; do
    echo "### fgprobe $shape $v 8 1 1 1   (B N reps variant cap xcd fused nt)"
    timeout -k 10 120 tools/bin/fgprobe $shape $v 8 1 1 1 | grep "per XCC\|us/launch"
  done
done
} > $O/fgprobe.txt 2>&1
echo "exit $?"; cat $O/fgprobe.txt
