#!/bin/bash
# round-2 experiment B: cooperative write shapes (wrbench modes 7-10), fused variants
mkdir -p gpurun_out/r02b
O=gpurun_out/r02b
W=tools/bin/wrbench
{
echo "== mode 4 reference points"; $W 4 4 0; $W 4 52 23400; $W 4 52 0
echo "== mode 7: GRP waves interleave KiB chunks of one S KiB region (lds caps workgroups per CU)"
for lds in 0 23400 40000 80000; do for grp in 2 4 8; do $W 7 52 $lds $grp; done; done
for lds in 0 40000; do for grp in 2 4; do $W 7 26 $lds $grp; $W 7 13 $lds $grp; done; done
echo "== mode 8: GRP waves, wave w writes the w-th contiguous piece"
for lds in 0 40000 80000; do for grp in 4 8; do $W 8 52 $lds $grp; done; done
echo "== mode 9: idle (delay x 64 s_sleep(8)) then S KiB"
for d in 0 2 8 32; do $W 9 4 0 1 $d; $W 9 4 23400 1 $d; done
for d in 2 8; do $W 9 52 23400 1 $d; done
echo "== mode 10: mode 7 + 1 KiB read + idle per wave"
for d in 0 2 8; do for grp in 4 8; do $W 10 52 40000 $grp $d; $W 10 52 0 $grp $d; done; done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
echo "== fgbench fused variants"
timeout -k 10 300 tools/bin/fgbench reps=40 \
  4096,200,64,7,0 4096,200,64,7,1 4096,200,64,7,2 4096,200,64,8,2 4096,200,64,0,2 4096,200,64,7,0 4096,200,64,7,2 \
  1024,200,64,0,0 1024,200,64,0,1 1024,200,64,0,2 1024,200,32,0,2 1024,200,40,0,2 \
  512,200,64,0,2 512,200,32,0,2 128,200,64,0,2 \
  400,2000,64,7,0 400,2000,64,7,2 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
