#!/bin/bash
# write-stream shapes, part 4: is it the LENGTH of a wave's stream or the compactness of the in-flight address window?
set -e
W=tools/bin/wrbench; O=gpurun_out/r02x; mkdir -p $O
{
echo "== reference points (mode 4)"
for S in 4 52; do timeout -k 5 60 $W 4 $S 23400; done
echo "== mode 13: short streams, segments dealt in scattered order"
for S in 4 8 52; do timeout -k 5 60 $W 13 $S 23400; done
echo "== mode 11: long-lived waves (52 KiB each), chunks interleaved over a super-group of G waves (grp = chunk KiB, delay = G, 0 = all)"
for G in 0 16384 2048 256 16; do for C in 1 4; do timeout -k 5 60 $W 11 52 23400 $C $G; done; done
for G in 0 2048; do timeout -k 5 60 $W 11 52 0 4 $G; done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
