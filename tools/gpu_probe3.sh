#!/bin/bash
for P in fgprobe2 fgprobe3; do
  for ipb in 1 2 4 8; do timeout -k 5 60 ./tools/bin/$P 4096 200 30 $ipb | head -1; done
done
timeout -k 5 60 ./tools/bin/fgprobe2 4096 200 30 4
timeout -k 5 60 ./tools/bin/fgprobe2 512 2000 30 4 | head -1
timeout -k 5 60 ./tools/bin/fgprobe2 1024 200 30 2 | head -1
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
