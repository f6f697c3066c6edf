#!/bin/bash
# write-stream shapes, part 9: 52 KiB per wave written as 4 KiB chunks interleaved over a small group of G waves (mode 11), G = 4 ... 64
W=tools/bin/wrbench; O=gpurun_out/r02aq; mkdir -p $O
{
for rep in 1 2 3; do
  timeout -k 5 60 $W 4 52 23400
  for G in 4 8 16 64; do timeout -k 5 60 $W 11 52 23400 4 $G; done
  timeout -k 5 60 $W 11 52 23400 2 4
  timeout -k 5 60 $W 11 52 23400 13 4
done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
