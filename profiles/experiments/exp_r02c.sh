#!/bin/bash
# round-2 experiment C: XCD placement of write streams; plain vs non-temporal stores when the outputs fit the Infinity Cache
mkdir -p gpurun_out/r02c
O=gpurun_out/r02c
W=tools/bin/wrbench
{
echo "== XCD-contiguous (mode 3) vs launch-order (mode 4)"
for S in 4 13 52; do for lds in 0 23400; do $W 4 $S $lds; $W 3 $S $lds; done; done
echo "== plain stores, launch order (mode 0)"
for S in 4 52; do $W 0 $S 0; $W 0 $S 23400; done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
for bin in fgbench fgbench_plain; do
echo "== $bin"
timeout -k 10 300 tools/bin/$bin reps=40 \
  4096,200,64,7,0 4096,200,64,0,0 2048,200,64,7,0 2048,200,64,0,0 \
  1024,200,64,0,0 1024,200,64,7,0 1024,200,32,0,0 512,200,64,0,0 512,200,32,0,0 256,200,64,0,0 128,200,64,0,0 \
  > $O/$bin.md 2>&1
cat $O/$bin.md
done
