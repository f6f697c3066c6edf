#!/bin/bash
# round-2 experiment K: compact pattern A/B (own vs library sin/cos, cap, xcd, fused)
mkdir -p gpurun_out/r02k
O=gpurun_out/r02k
for bin in fgbench fgbench_libsc; do
echo "== $bin"
timeout -k 10 300 tools/bin/$bin reps=40 pat=1 nt=1 xcd=0 4096,200,64,8,0 4096,200,64,0,0 xcd=1 4096,200,64,8,1 4096,200,64,0,1 4096,200,64,12,1 pat=0 xcd=1 4096,200,64,8,1 > $O/$bin.md 2>&1
cat $O/$bin.md
done
