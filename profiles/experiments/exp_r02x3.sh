#!/bin/bash
# write-stream shapes, part 6: cache-policy bits of the 16-byte stores
W=tools/bin/wrbench; O=gpurun_out/r02x3; mkdir -p $O
{
for S in 52 4; do
  echo "== S = $S KiB per wave; policy 0 none, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 nt, 6 sc1 nt, 7 sc0 sc1 nt"
  timeout -k 5 60 $W 4 $S 23400
  for P in 0 1 2 3 4 5 6 7; do timeout -k 5 60 $W 15 $S 23400 $P; done
done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
