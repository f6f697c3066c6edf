#!/bin/bash
for r in 1 2 3; do
timeout -k 5 60 ./tools/bin/fgprobe 4096 200 30 1 | head -1
timeout -k 5 60 ./tools/bin/fgprobe_nt 4096 200 30 1 | head -1
done
timeout -k 5 60 ./tools/bin/fgprobe 512 2000 30 1 | head -1
timeout -k 5 60 ./tools/bin/fgprobe_nt 512 2000 30 1 | head -1
