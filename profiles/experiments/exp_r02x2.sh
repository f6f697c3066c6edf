#!/bin/bash
# write-stream shapes, part 5: one store instruction spread over several places of the wave's region
W=tools/bin/wrbench; O=gpurun_out/r02x2; mkdir -p $O
{
echo "== reference points (mode 4)"
for S in 4 52; do timeout -k 5 60 $W 4 $S 23400; done
echo "== mode 14: grp = lanes per contiguous piece (64 = mode 4)"
for S in 52 48 32 16; do for LG in 64 32 16 8 4 2; do timeout -k 5 60 $W 14 $S 23400 $LG; done; done
timeout -k 5 60 $W 4 52 23400
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
