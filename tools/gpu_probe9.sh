#!/bin/bash
W=./tools/bin/wrbench
for r in 1 2; do
for S in 4 16 43 52; do timeout -k 5 60 $W 0 $S 17920; timeout -k 5 60 $W 4 $S 17920; done
timeout -k 5 60 $W 3 43 17920
timeout -k 5 60 $W 5 43 17920 16
timeout -k 5 60 $W 5 43 17920 4
done
